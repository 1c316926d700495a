// dist-primitive/src/dsumcheck.rs: the Phase-1 loops routed through libzkhip.so -- every function keeps the reference's
// signature.  UNCOMPILED here (no Rust toolchain).  Only the replaced parts are written out; "unchanged" names the reference
// lines that stay as they are (they are not reproduced in this repository).
use crate::unpack::pss2ss;
use crate::utils::serializing_net::MPCSerializeNet;
use crate::zkhip_party::{check, is_bls12_381_fr, ZkParty};
use crate::zkhip_sys::*;
use ark_ff::FftField;
use mpc_net::{MPCNetError, MultiplexedStreamID};
use secret_sharing::pss::PackedSharingParams;

/// the n = log2(len) rounds (sum_lo, sum_hi) + fold of dsumcheck.rs:10-21 = :107-121 = :301-315 on the GPU
fn gpu_sumcheck_rounds<F: FftField>(party: &ZkParty, table: &[F], challenge: &[F]) -> (Vec<(F, F)>, F) {
    let n = table.len().trailing_zeros() as usize;
    let d = party.upload(table).unwrap();
    let mut pairs = vec![F::zero(); 2 * n];
    let mut last = F::zero();
    check(party.ctx, unsafe {
        zk_sumcheck(party.ctx, d.ptr, table.len(), challenge.as_ptr() as *const u64, pairs.as_mut_ptr() as *mut u64, &mut last as *mut F as *mut u64)
    }).unwrap();
    (pairs.chunks(2).map(|p| (p[0], p[1])).collect(), last)
}
/// the n rounds (t0, t1, t2) + fold of both tables, dsumcheck.rs:37-85 = :167-219 = :377-429
fn gpu_product_rounds<F: FftField>(party: &ZkParty, f: &[F], g: &[F], challenge: &[F]) -> (Vec<(F, F, F)>, F, F) {
    let n = f.len().trailing_zeros() as usize;
    let (df, dg) = (party.upload(f).unwrap(), party.upload(g).unwrap());
    let mut triples = vec![F::zero(); 3 * n];
    let (mut lf, mut lg) = (F::zero(), F::zero());
    check(party.ctx, unsafe {
        zk_sumcheck_product(party.ctx, df.ptr, dg.ptr, f.len(), challenge.as_ptr() as *const u64, triples.as_mut_ptr() as *mut u64,
                            &mut lf as *mut F as *mut u64, &mut lg as *mut F as *mut u64)
    }).unwrap();
    (triples.chunks(3).map(|t| (t[0], t[1], t[2])).collect(), lf, lg)
}

pub fn sumcheck<F: FftField>(evaluation: &Vec<F>, challenge: &Vec<F>) -> Vec<(F, F)> {
    if let (true, Some(party)) = (is_bls12_381_fr::<F>(), ZkParty::any()) {
        let (mut result, last) = gpu_sumcheck_rounds(&party, evaluation, challenge);
        result.push((F::ZERO, last)); // :24
        return result;
    }
    sumcheck_cpu(evaluation, challenge) // dsumcheck.rs:7-25 unchanged
}

pub fn sumcheck_product<F: FftField>(evaluation_f: &Vec<F>, evaluation_g: &Vec<F>, challenge: &Vec<F>) -> Vec<(F, F, F)> {
    assert_eq!(evaluation_f.len(), evaluation_g.len());
    if let (true, Some(party)) = (is_bls12_381_fr::<F>(), ZkParty::any()) {
        let (mut result, lf, lg) = gpu_product_rounds(&party, evaluation_f, evaluation_g, challenge);
        result.push((F::ZERO, lf * lg, F::ZERO)); // :88
        return result;
    }
    sumcheck_product_cpu(evaluation_f, evaluation_g, challenge) // :33-89 unchanged
}

pub async fn c_sumcheck<F: FftField, Net: MPCSerializeNet>(
    shares: &Vec<F>, challenge: &Vec<F>, pp: &PackedSharingParams<F>, net: &Net, sid: MultiplexedStreamID,
) -> Result<Vec<(F, F)>, MPCNetError> {
    let (mut result, last) = match (is_bls12_381_fr::<F>(), ZkParty::of(net)) {
        (true, Some(party)) => gpu_sumcheck_rounds(&party, shares, challenge), // replaces the loop :107-121
        _ => return c_sumcheck_cpu(shares, challenge, pp, net, sid).await,
    };
    let mut last_round = pss2ss(last, pp, net, sid).await?; // :125
    // Phase 2 on the l-vector, re-using challenge[0..log2 l] (:129), and the closing (0, last) row: :126-145 unchanged,
    // starting from `result` and `last_round`
    c_sumcheck_phase2(&mut result, &mut last_round, challenge, pp);
    Ok(result)
}

pub async fn c_sumcheck_product<F: FftField, Net: MPCSerializeNet>(
    shares_f: &Vec<F>, shares_g: &Vec<F>, challenge: &Vec<F>, pp: &PackedSharingParams<F>, net: &Net, sid: MultiplexedStreamID,
) -> Result<Vec<(F, F, F)>, MPCNetError> {
    assert_eq!(shares_f.len(), shares_g.len()); // :160
    let (mut result, lf, lg) = match (is_bls12_381_fr::<F>(), ZkParty::of(net)) {
        (true, Some(party)) => gpu_product_rounds(&party, shares_f, shares_g, challenge), // replaces the loop :167-219
        _ => return c_sumcheck_product_cpu(shares_f, shares_g, challenge, pp, net, sid).await,
    };
    let mut last_round_f = pss2ss(lf, pp, net, sid).await?; // :224
    let mut last_round_g = pss2ss(lg, pp, net, sid).await?; // :225
    // Phase 2 (:226-281) and the closing row (0, f*g, 0) (:282): unchanged
    c_sumcheck_product_phase2(&mut result, &mut last_round_f, &mut last_round_g, challenge, pp);
    Ok(result)
}

pub async fn d_sumcheck<F: FftField, Net: MPCSerializeNet>(
    partial_poly: &Vec<F>, challenge: &Vec<F>, net: &Net, sid: MultiplexedStreamID,
) -> Result<Vec<(F, F)>, MPCNetError> {
    let (mut result, last) = match (is_bls12_381_fr::<F>(), ZkParty::of(net)) {
        (true, Some(party)) => gpu_sumcheck_rounds(&party, partial_poly, challenge), // replaces the loop :301-315
        _ => return d_sumcheck_cpu(partial_poly, challenge, net, sid).await,
    };
    result.push((F::ZERO, last)); // :318
    // gather of the round vectors and the leader's s rounds over the parties' last values: :320-356 unchanged
    d_sumcheck_leader(result, challenge, net, sid).await
}

pub async fn d_sumcheck_product<F: FftField, Net: MPCSerializeNet>(
    partial_f: &Vec<F>, partial_g: &Vec<F>, challenge: &Vec<F>, net: &Net, sid: MultiplexedStreamID,
) -> Result<Vec<(F, F, F)>, MPCNetError> {
    assert_eq!(partial_f.len(), partial_g.len()); // :370
    let (mut result, lf, lg) = match (is_bls12_381_fr::<F>(), ZkParty::of(net)) {
        (true, Some(party)) => gpu_product_rounds(&party, partial_f, partial_g, challenge), // replaces the loop :377-429
        _ => return d_sumcheck_product_cpu(partial_f, partial_g, challenge, net, sid).await,
    };
    result.push((lg, lf, F::ZERO)); // the marker tuple is (g, f, 0)  :433
    // gather (:437), party-wise sums (:440-447), s leader rounds with f from .1 and g from .0 (:448-507): unchanged
    d_sumcheck_product_leader(result, challenge, net, sid).await
}
// `*_cpu`, `*_phase2`, `*_leader`: the reference's own statements at the cited lines, moved into helpers (not reproduced here).
