// dist-primitive/src/dmsm.rs with the hot path routed through libzkhip.so (MI355X).
// UNCOMPILED in this repository (the build image has no Rust toolchain); it documents, line by line,
// what replaces dist-primitive/src/dmsm.rs:9-43.  Symbols: rust/zkhip_sys.rs (generated from include/zkhip.h).
//
// What changes against the reference:
//   :19-24  `G::msm(b, s).unwrap()` per batch item      -> ONE zk_d_msm call for the whole batch
//   :29-40  `leader_compute_element` (gather to party 0 over TCP, unpack2 -> sum -> pack_from_public on the
//           leader, scatter)                             -> inside zk_d_msm: an RCCL all-gather of the 144-byte
//           results over xGMI and the same PUBLIC linear map evaluated by every party for its own slot
//   bases   `&Vec<Vec<G::Affine>>` cloned per call (dpoly_comm.rs:258) -> SRS levels registered once (`ZkSrs`)
//   scalars `&Vec<Vec<Fr>>` on the host                  -> device-resident share tables (`DeviceFr`)
// The signature keeps the reference's shape; `Net` shrinks to the handle that owns the communicator.
use crate::zkhip_sys::*;
use ark_bls12_381::{Fr, G1Projective};
use ark_ff::{BigInteger, PrimeField};
use secret_sharing::pss::PackedSharingParams;

/// one GPU = one party: owns the context handle (stream, scratch, RCCL communicator)
pub struct ZkParty {
    pub ctx: *mut ZkCtx,
    pub party_id: usize,
    pub n_parties: usize,
}

/// a resident table of Fr shares (`zk_malloc` + `zk_memcpy_h2d`, Montgomery limbs exactly as `Vec<Fr>` holds them)
pub struct DeviceFr {
    pub ptr: *mut std::os::raw::c_void,
    pub len: usize,
}

#[derive(Debug)]
pub enum ZkError {
    Invalid, Length(usize), Hip, NoDevice, DivZero, Oom, Comm(String),
}

fn check(ctx: *mut ZkCtx, rc: i32) -> Result<(), ZkError> {
    match rc {
        0 => Ok(()),
        ZK_ERR_LENGTH => Err(ZkError::Length(0)), // `msm` -> Err(min_len): the reference unwrap()s it (dmsm.rs:23)
        ZK_ERR_COMM => Err(ZkError::Comm(unsafe { std::ffi::CStr::from_ptr(zk_last_error(ctx)) }.to_string_lossy().into_owned())), // MPCNetError
        ZK_ERR_OOM => Err(ZkError::Oom),
        ZK_ERR_HIP => Err(ZkError::Hip),
        _ => Err(ZkError::Invalid),
    }
}

fn canonical(x: Fr) -> [u64; 4] {
    let b = x.into_bigint();
    let mut out = [0u64; 4];
    out.copy_from_slice(b.as_ref());
    out
}

/// This protocol implements dMSM in a batched way (dist-primitive/src/dmsm.rs:9-43).
pub fn d_msm(
    bases: &[*const ZkSrs],          // powers_of_g[level] handles, one per batch item
    scalars: &[DeviceFr],            // this party's packed scalar shares, resident in HBM
    pp: &PackedSharingParams<Fr>,
    party: &ZkParty,
) -> Result<Vec<G1Projective>, ZkError> {
    assert_eq!(bases.len(), scalars.len()); // dmsm.rs:16
    let count = bases.len();
    // the leader closure of :30-39 as coefficients: out_p = c_p * sum_i lambda_i * C_i with
    //   lambda_i = sum_j unpack2[j][i],  c_p = sum_j pack[p][j]   (any packing factor l).
    // lambda_p is folded into this party's scalars on the device (MSM(b, lambda s) = lambda MSM(b, s)),
    // so every coefficient of the point combination is the same c_p: 7 additions + one scalar multiplication.
    let n = party.n_parties;
    let mut unit = vec![Fr::from(0u64); n];
    unit[party.party_id] = Fr::from(1u64);
    let lambda: Fr = pp.unpack2(unit).iter().sum();
    let c_p: Fr = pp.pack_from_public(vec![Fr::from(1u64); pp.l])[party.party_id];
    let lambda_mont: [u64; 4] = lambda.0 .0; // Montgomery limbs, as stored
    let coeffs: Vec<[u64; 4]> = vec![canonical(c_p); n];

    let ptrs: Vec<*const std::os::raw::c_void> = scalars.iter().map(|s| s.ptr as *const _).collect();
    let lens: Vec<usize> = scalars.iter().map(|s| s.len).collect();
    let mut out = vec![[0u64; 18]; count];
    check(party.ctx, unsafe {
        zk_d_msm(party.ctx, count, bases.as_ptr(), std::ptr::null(), ptrs.as_ptr(), lens.as_ptr(),
                 lambda_mont.as_ptr(), coeffs.as_ptr() as *const u64, out.as_mut_ptr() as *mut u64)
    })?;
    // results are normalised Jacobian points in arkworks' own layout ({x, y, z}: 3 x [u64; 6], Montgomery):
    // valid `G1Projective` values, equal to the reference's result as group elements
    Ok(out.into_iter().map(|o| unsafe { std::mem::transmute_copy::<[u64; 18], G1Projective>(&o) }).collect())
}
