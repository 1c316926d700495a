// dist-primitive/src/dpoly_comm.rs: commit / open / c_open with their MSMs and fold loops on the GPU; signatures unchanged
// (`impl<E: Pairing> PolynomialCommitment<E>`).  UNCOMPILED here (no Rust toolchain).  `c_commit` (:244-267) and `d_commit`
// (:276-297) need NO edit: they call `d_msm` / `d_local_commit`, which are patched.
use crate::zkhip_party::{check, is_bls12_381_g1, ZkParty};
use crate::zkhip_sys::*;
use std::os::raw::c_void;

impl<E: Pairing> PolynomialCommitment<E> {
    /// :237-243 (= d_local_commit :269-275)
    pub fn commit(&self, peval: &Vec<E::ScalarField>) -> E::G1 {
        let level = peval.len().trailing_zeros() as usize;
        assert!(level < self.powers_of_g.len());
        assert!(peval.len() == 2_usize.pow(level as u32));
        if let (true, Some(party)) = (is_bls12_381_g1::<E::G1>(), ZkParty::any()) {
            let srs = party.srs_of(&self.powers_of_g[level]).unwrap();
            let d = party.upload(peval).unwrap();
            let mut out = [0u64; 18];
            check(party.ctx, unsafe { zk_msm_g1(party.ctx, srs, 0, d.ptr, peval.len(), out.as_mut_ptr()) }).unwrap();
            return unsafe { std::mem::transmute_copy::<[u64; 18], E::G1>(&out) };
        }
        E::G1::msm(&self.powers_of_g[level], peval).unwrap()
    }

    /// :299-325 (= d_local_open :327-353): the n rounds q = hi - lo, fold, commit(q) become ONE zk_open_rounds and ONE
    /// zk_msm_g1_batch over the n slices of the quotient buffer (q_0 at offset 0 with len/2 elements, q_1 after it, ...)
    pub fn open(&self, peval: &Vec<E::ScalarField>, point: &[E::ScalarField]) -> (E::ScalarField, Vec<E::G1>) {
        let n = peval.len().trailing_zeros() as usize;
        assert_eq!(peval.len(), 2_usize.pow(n as u32));
        let party = match (is_bls12_381_g1::<E::G1>(), ZkParty::any()) {
            (true, Some(p)) => p,
            _ => return self.open_cpu(peval, point), // :305-324 unchanged
        };
        let d = party.upload(peval).unwrap();
        let q = party.alloc_fr(peval.len().max(2) - 1).unwrap();
        let mut value = E::ScalarField::zero();
        check(party.ctx, unsafe {
            zk_open_rounds(party.ctx, d.ptr, peval.len(), point.as_ptr() as *const u64, q.ptr, &mut value as *mut _ as *mut u64)
        }).unwrap();
        (value, self.commit_quotients(&party, q.ptr, peval.len()))
    }

    /// the n commitments `commit(q_i)` of one open: levels n-1 .. 0 of powers_of_g, scalars = slices of the device buffer
    fn commit_quotients(&self, party: &ZkParty, d_q: *mut c_void, len: usize) -> Vec<E::G1> {
        let n = len.trailing_zeros() as usize;
        let (mut srs, mut ptrs, mut lens, mut off, mut m) = (Vec::new(), Vec::new(), Vec::new(), 0usize, len);
        for _ in 0..n {
            let h = m / 2;
            srs.push(party.srs_of(&self.powers_of_g[h.trailing_zeros() as usize]).unwrap());
            ptrs.push(unsafe { (d_q as *const u8).add(32 * off) } as *const c_void);
            lens.push(h);
            off += h;
            m = h;
        }
        let mut out = vec![[0u64; 18]; n];
        check(party.ctx, unsafe { zk_msm_g1_batch(party.ctx, n, srs.as_ptr(), std::ptr::null(), ptrs.as_ptr(), lens.as_ptr(), out.as_mut_ptr() as *mut u64) }).unwrap();
        out.iter().map(|o| unsafe { std::mem::transmute_copy::<[u64; 18], E::G1>(o) }).collect()
    }

    /// :401-464.  Phase 1 (:418-432) on the GPU; `self.c_commit(&result, ..)` (:436) is replaced by a d_msm over the device
    /// slices (no download / re-upload of the q_i); pss2ss and Phase 2 (:439-462) unchanged.
    pub async fn c_open<Net: MPCSerializeNet>(
        &self, peval: &Vec<E::ScalarField>, point: &Vec<E::ScalarField>, pp: &PackedSharingParams<E::ScalarField>, net: &Net, sid: MultiplexedStreamID,
    ) -> Result<(E::ScalarField, Vec<E::G1>), MPCNetError> {
        let n: usize = peval.len().trailing_zeros() as usize;
        assert_eq!(peval.len(), 2_usize.pow(n as u32));
        let party = match (is_bls12_381_g1::<E::G1>(), ZkParty::of(net)) {
            (true, Some(p)) => p,
            _ => return self.c_open_cpu(peval, point, pp, net, sid).await,
        };
        let d = party.upload(peval)?;
        let q = party.alloc_fr(peval.len().max(2) - 1)?;
        let mut last = E::ScalarField::zero();
        check(party.ctx, unsafe {
            zk_open_rounds(party.ctx, d.ptr, peval.len(), point.as_ptr() as *const u64, q.ptr, &mut last as *mut _ as *mut u64)
        })?;
        // the batched commitment of all q_i (:436): bases_i = powers_of_g[log2(|q_i| * l)] (:250-258), through zk_d_msm
        let mut res = self.c_commit_device(&party, q.ptr, peval.len(), pp, net, sid).await?;
        let mut current_r = pss2ss(last, pp, net, sid).await?; // :439
        assert!(current_r.len() == pp.l);
        // Phase 2 on the l-vector re-using point[0..] (:452), each round pushing a local commit(q): :441-462 unchanged
        let value = self.c_open_phase2(&mut res, &mut current_r, point, pp);
        Ok((value, res))
    }
}
// `open_cpu`, `c_open_cpu`, `c_open_phase2`: the reference's statements at the cited lines moved into helpers (not reproduced).
// `c_commit_device`: d_msm_patched's body with the scalar pointers taken from the device buffer instead of `party.upload`.
