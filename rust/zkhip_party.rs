// dist-primitive/src/zkhip_party.rs (NEW file in the reference tree) -- the glue that lets the patched functions keep the
// reference's exact signatures: a per-party GPU context looked up from the `Net` value the caller already passes, an
// SRS-handle cache keyed on the identity of the `Vec<G::Affine>` the caller passes, host <-> HBM staging of `&Vec<F>`.
// UNCOMPILED in this repository (no Rust toolchain in the build image); symbols: rust/zkhip_sys.rs (generated from
// include/zkhip.h, kept in sync by tests/test_abi.py).
//
// Ownership model.  One `ZkParty` per party (= per GPU).  In the reference's local mode every party is a tokio task of one
// process (mpc-net/src/multi.rs:330-352): `ZkParty::init_all(n)` creates a ctx per GPU and one RCCL communicator over all
// of them (zk_comm_init_all).  In the multi-process mode each process calls `ZkParty::init(party_id, n, id)` with the
// 128-byte id party 0 obtained from zk_comm_unique_id and sent over the existing mpc-net channel.  Without a registered
// party every patched function falls through to the reference's own CPU code, so the crate still works without a GPU box.
use crate::zkhip_sys::*;
use ark_ff::PrimeField;
use mpc_net::{MPCNet, MPCNetError};
use std::any::TypeId;
use std::collections::HashMap;
use std::os::raw::c_void;
use std::sync::{Arc, Mutex, OnceLock};

pub struct ZkParty {
    pub ctx: *mut ZkCtx,
    pub party_id: usize,
    pub n_parties: usize,
    pub has_comm: bool,
    /// (data pointer, length) of a `Vec<G1Affine>` the prover passed -> the registered (and window-tabled) device copy.
    /// `PolynomialCommitment::powers_of_g` lives as long as the prover, and `c_commit` clones the level per call
    /// (dpoly_comm.rs:258), so the clone's CONTENT is what identifies it: the key also carries a 64-bit fingerprint of the
    /// first, middle and last point; a hit costs three 104-byte reads instead of a 100 MB upload.
    srs: Mutex<HashMap<(usize, [u64; 3]), usize>>,
}
unsafe impl Send for ZkParty {}
unsafe impl Sync for ZkParty {}

static PARTIES: OnceLock<Mutex<HashMap<usize, Arc<ZkParty>>>> = OnceLock::new();
fn parties() -> &'static Mutex<HashMap<usize, Arc<ZkParty>>> {
    PARTIES.get_or_init(|| Mutex::new(HashMap::new()))
}

#[derive(Debug)]
pub struct ZkError(pub i32, pub String);
impl From<ZkError> for MPCNetError {
    fn from(e: ZkError) -> Self {
        // ZK_ERR_COMM is the reference's MPCNetError; everything else is a panic in the reference (unwrap / assert!)
        if e.0 == ZK_ERR_COMM { MPCNetError::Generic(e.1) } else { panic!("zkhip error {}: {}", e.0, e.1) }
    }
}
pub fn check(ctx: *mut ZkCtx, rc: i32) -> Result<(), ZkError> {
    if rc == 0 { return Ok(()); }
    let msg = unsafe { std::ffi::CStr::from_ptr(zk_last_error(ctx)) }.to_string_lossy().into_owned();
    Err(ZkError(rc, msg))
}

impl ZkParty {
    /// local mode: one process, party p on GPU p (mpc-net/src/multi.rs:330-352)
    pub fn init_all(n_parties: usize) -> Result<(), ZkError> {
        assert!(unsafe { zk_device_count() } as usize >= n_parties, "one GPU per party");
        let mut ctxs: Vec<*mut ZkCtx> = vec![std::ptr::null_mut(); n_parties];
        for (p, c) in ctxs.iter_mut().enumerate() {
            check(std::ptr::null_mut(), unsafe { zk_ctx_create(p as i32, c) })?;
        }
        check(ctxs[0], unsafe { zk_comm_init_all(ctxs.as_ptr() as *const *mut ZkCtx, n_parties as i32) })?;
        let mut reg = parties().lock().unwrap();
        for (p, &ctx) in ctxs.iter().enumerate() {
            reg.insert(p, Arc::new(ZkParty { ctx, party_id: p, n_parties, has_comm: true, srs: Mutex::new(HashMap::new()) }));
        }
        Ok(())
    }
    /// one process per party: `id` = zk_comm_unique_id of party 0, handed over the existing mpc-net channel
    pub fn init(party_id: usize, n_parties: usize, device: i32, id: Option<&[u8; ZK_COMM_ID_BYTES]>) -> Result<(), ZkError> {
        let mut ctx = std::ptr::null_mut();
        check(std::ptr::null_mut(), unsafe { zk_ctx_create(device, &mut ctx) })?;
        if let Some(id) = id {
            check(ctx, unsafe { zk_comm_init(ctx, party_id as i32, n_parties as i32, id.as_ptr()) })?;
        }
        parties().lock().unwrap().insert(party_id, Arc::new(ZkParty { ctx, party_id, n_parties, has_comm: id.is_some(), srs: Mutex::new(HashMap::new()) }));
        Ok(())
    }
    /// the party context of the `net` a dist-primitive function was called with (None: stay on the CPU path)
    pub fn of<Net: MPCNet>(net: &Net) -> Option<Arc<ZkParty>> {
        parties().lock().unwrap().get(&(net.party_id() as usize)).cloned()
    }
    /// any registered party (the non-distributed functions `commit`, `open`, `sumcheck`, `acc_product` take no `net`)
    pub fn any() -> Option<Arc<ZkParty>> {
        parties().lock().unwrap().values().next().cloned()
    }

    /// `&Vec<G1Affine>` -> resident SRS handle (registered + window table on first sight)
    pub fn srs_of<A: 'static>(&self, bases: &[A]) -> Result<*const ZkSrs, ZkError> {
        assert_eq!(TypeId::of::<A>(), TypeId::of::<ark_bls12_381::G1Affine>());
        let stride = std::mem::size_of::<A>(); // 104: { x: [u64; 6], y: [u64; 6], infinity: bool }
        let words = |i: usize| -> u64 {
            let p = unsafe { std::slice::from_raw_parts((bases.as_ptr() as *const u8).add(i * stride) as *const u64, 12) };
            p.iter().fold(0xcbf29ce484222325u64, |h, w| (h ^ w).wrapping_mul(0x100000001b3))
        };
        let n = bases.len();
        let key = (n, if n == 0 { [0; 3] } else { [words(0), words(n / 2), words(n - 1)] });
        let mut cache = self.srs.lock().unwrap();
        if let Some(&h) = cache.get(&key) { return Ok(h as *const ZkSrs); }
        let mut h: *mut ZkSrs = std::ptr::null_mut();
        check(self.ctx, unsafe { zk_srs_register(self.ctx, bases.as_ptr() as *const c_void, stride, n, &mut h) })?;
        if n >= 64 && n <= (1 << 22) { check(self.ctx, unsafe { zk_srs_precompute(self.ctx, h, 0) })?; }
        cache.insert(key, h as usize);
        Ok(h as *const ZkSrs)
    }

    /// `&Vec<F>` (Montgomery limbs exactly as arkworks stores them) -> a device table; freed by `DeviceFr::drop`
    pub fn upload<F: PrimeField>(&self, v: &[F]) -> Result<DeviceFr, ZkError> {
        assert_eq!(std::mem::size_of::<F>(), 32);
        let mut ptr: *mut c_void = std::ptr::null_mut();
        check(self.ctx, unsafe { zk_malloc(self.ctx, v.len().max(1) * 32, &mut ptr) })?;
        if !v.is_empty() { check(self.ctx, unsafe { zk_memcpy_h2d(self.ctx, ptr, v.as_ptr() as *const c_void, v.len() * 32) })?; }
        Ok(DeviceFr { ctx: self.ctx, ptr, len: v.len() })
    }
    pub fn alloc_fr(&self, len: usize) -> Result<DeviceFr, ZkError> {
        let mut ptr: *mut c_void = std::ptr::null_mut();
        check(self.ctx, unsafe { zk_malloc(self.ctx, len.max(1) * 32, &mut ptr) })?;
        Ok(DeviceFr { ctx: self.ctx, ptr, len })
    }
}

pub struct DeviceFr {
    ctx: *mut ZkCtx,
    pub ptr: *mut c_void,
    pub len: usize,
}
impl DeviceFr {
    pub fn download<F: PrimeField>(&self, offset: usize, len: usize) -> Result<Vec<F>, ZkError> {
        let mut out = vec![F::zero(); len];
        check(self.ctx, unsafe { zk_memcpy_d2h(self.ctx, out.as_mut_ptr() as *mut c_void, (self.ptr as *const u8).add(offset * 32) as *const c_void, len * 32) })?;
        Ok(out)
    }
}
impl Drop for DeviceFr {
    fn drop(&mut self) { unsafe { zk_free(self.ctx, self.ptr) }; } // parks the block, no device synchronisation
}

/// is `G` the curve the library accelerates?  (the reference's own tests also instantiate BLS12-377: CPU path)
pub fn is_bls12_381_g1<G: 'static>() -> bool { TypeId::of::<G>() == TypeId::of::<ark_bls12_381::G1Projective>() }
pub fn is_bls12_381_fr<F: 'static>() -> bool { TypeId::of::<F>() == TypeId::of::<ark_bls12_381::Fr>() }
