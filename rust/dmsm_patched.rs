// dist-primitive/src/dmsm.rs with the hot path routed through libzkhip.so (MI355X) -- SAME SIGNATURE as the reference
// (dmsm.rs:9-15), so `c_commit` (dpoly_comm.rs:244-267), `c_open` (:401-464) and every caller in hyperplonk/src/dhyperplonk.rs
// compile against it unchanged.  UNCOMPILED in this repository (no Rust toolchain in the build image).
//
// What changes inside the function:
//   :19-24  `G::msm(b, s).unwrap()` per batch item -> the bases are looked up in the party's SRS cache (registered once,
//           window table built once: zkhip_party.rs `srs_of`), the scalar vectors are uploaded, ONE call runs the batch
//   :29-40  on one node (a communicator is attached to the party): the gather / leader closure / scatter is zk_d_msm's RCCL
//           all-gather + the same public map on every party; across machines the reference's `leader_compute_element`
//           stays, fed with the GPU's local results
//   anything else (another curve, no GPU party registered): the reference's own code path
use crate::utils::{operator::transpose, serializing_net::MPCSerializeNet};
use crate::zkhip_party::{check, is_bls12_381_g1, ZkParty};
use crate::zkhip_sys::*;
use ark_ec::CurveGroup;
use ark_ff::{BigInteger, PrimeField};
use mpc_net::{MPCNetError, MultiplexedStreamID};
use secret_sharing::pss::PackedSharingParams;
use std::os::raw::c_void;

fn canonical<F: PrimeField>(x: F) -> [u64; 4] {
    let b = x.into_bigint();
    let mut out = [0u64; 4];
    out.copy_from_slice(b.as_ref());
    out
}

/// This protocol implement dMSM in a batched way.        (dist-primitive/src/dmsm.rs:9-43)
pub async fn d_msm<G: CurveGroup, Net: MPCSerializeNet>(
    bases: &Vec<Vec<G::Affine>>,
    scalars: &Vec<Vec<G::ScalarField>>,
    pp: &PackedSharingParams<G::ScalarField>,
    net: &Net,
    sid: MultiplexedStreamID,
) -> Result<Vec<G>, MPCNetError> {
    assert_eq!(bases.len(), scalars.len()); // :16
    let party = match (is_bls12_381_g1::<G>(), ZkParty::of(net)) {
        (true, Some(p)) => p,
        _ => return d_msm_cpu(bases, scalars, pp, net, sid).await, // the reference body, dmsm.rs:17-42, unchanged
    };
    let count = bases.len();
    // `msm` -> Err(min_len), unwrap()ed at :23: same panic, before any device work
    for (b, s) in bases.iter().zip(scalars.iter()) {
        if b.len() != s.len() { Err::<G, usize>(b.len().min(s.len())).unwrap(); }
    }
    let srs: Vec<*const ZkSrs> = bases.iter().map(|b| party.srs_of(b)).collect::<Result<_, _>>()?;
    let dev: Vec<_> = scalars.iter().map(|s| party.upload(s)).collect::<Result<Vec<_>, _>>()?;
    let ptrs: Vec<*const c_void> = dev.iter().map(|d| d.ptr as *const c_void).collect();
    let lens: Vec<usize> = scalars.iter().map(|s| s.len()).collect();
    let mut out = vec![[0u64; 18]; count];
    if party.has_comm {
        // leader closure :30-39 as coefficients: out_p = c_p * sum_i lambda_i * C_i with lambda_i = sum_j unpack2[j][i] and
        // c_p = sum_j pack[p][j] (any packing factor l).  lambda_p goes into this party's scalars on the device
        // (MSM(b, lambda s) = lambda MSM(b, s)), every point coefficient is then the same c_p.
        let n = net.n_parties();
        let me = net.party_id() as usize;
        let mut unit = vec![G::ScalarField::from(0u64); n];
        unit[me] = G::ScalarField::from(1u64);
        let lambda: G::ScalarField = pp.unpack2(unit).iter().sum();
        let c_p: G::ScalarField = pp.pack_from_public(vec![G::ScalarField::from(1u64); pp.l])[me];
        let lambda_mont: [u64; 4] = unsafe { std::mem::transmute_copy(&lambda) }; // Montgomery limbs, as stored
        let coeffs: Vec<[u64; 4]> = vec![canonical(c_p); n];
        check(party.ctx, unsafe {
            zk_d_msm(party.ctx, count, srs.as_ptr(), std::ptr::null(), ptrs.as_ptr(), lens.as_ptr(), lambda_mont.as_ptr(),
                     coeffs.as_ptr() as *const u64, out.as_mut_ptr() as *mut u64)
        })?;
        net.add_comm(144 * count * (n - 1), 144 * count * (n - 1)); // what the all-gather moved, for get_comm()
        // normalised Jacobian {x, y, z} in arkworks' own layout: valid `Projective` values, equal to the reference's as group elements
        return Ok(out.iter().map(|o| unsafe { std::mem::transmute_copy::<[u64; 18], G>(o) }).collect());
    }
    // no on-node communicator: local MSMs on the GPU, the exchange as in the reference (:29-40)
    check(party.ctx, unsafe { zk_msm_g1_batch(party.ctx, count, srs.as_ptr(), std::ptr::null(), ptrs.as_ptr(), lens.as_ptr(), out.as_mut_ptr() as *mut u64) })?;
    let c_shares: Vec<G> = out.iter().map(|o| unsafe { std::mem::transmute_copy::<[u64; 18], G>(o) }).collect();
    net.leader_compute_element(&c_shares, sid, |shares| {
        let shares = transpose(shares);
        let results = shares.iter().map(|s| {
            let output = pp.unpack2(s.clone()).iter().sum();
            pp.pack_from_public(vec![output; pp.l])
        }).collect();
        transpose(results)
    }, "MSM Leader").await
}
// `d_msm_cpu` = the body of dist-primitive/src/dmsm.rs:16-42 moved into a function of the same signature (not reproduced here).
