// dist-primitive/src/dacc_product.rs: the product trees on the GPU; signatures unchanged.  UNCOMPILED here (no Rust toolchain).
use crate::utils::serializing_net::MPCSerializeNet;
use crate::zkhip_party::{check, is_bls12_381_fr, ZkParty};
use crate::zkhip_sys::*;
use ark_ff::FftField;
use mpc_net::{MPCNetError, MultiplexedStreamID};

/// tree = x || x; tree[N + j] = tree[2j] * tree[2j + 1]; tree[2N - 1] = 0     (dacc_product.rs:31-38 = :304-313 = :372-381)
fn gpu_subtree<F: FftField>(party: &ZkParty, inputs: &[F]) -> Vec<F> {
    let d = party.upload(inputs).unwrap();
    let t = party.alloc_fr(2 * inputs.len()).unwrap();
    check(party.ctx, unsafe { zk_product_tree(party.ctx, d.ptr, inputs.len(), t.ptr) }).unwrap();
    t.download(0, 2 * inputs.len()).unwrap()
}

pub fn acc_product<F: FftField>(x: &Vec<F>) -> (Vec<F>, Vec<F>, Vec<F>) {
    let tree = match (is_bls12_381_fr::<F>(), ZkParty::any()) {
        (true, Some(party)) => gpu_subtree(&party, x),
        _ => return acc_product_cpu(x), // :31-56 unchanged
    };
    // the three strided views (:41-55): v(x,0) = tree[0::2], v(x,1) = tree[1::2], v(1,x) = tree[N..]
    (tree.iter().step_by(2).cloned().collect(), tree.iter().skip(1).step_by(2).cloned().collect(), tree[x.len()..].to_vec())
}

pub async fn d_acc_product<F: FftField, Net: MPCSerializeNet>(
    inputs: &Vec<F>, net: &Net, sid: MultiplexedStreamID,
) -> Result<(Vec<F>, Option<Vec<F>>), MPCNetError> {
    let subtree = match (is_bls12_381_fr::<F>(), ZkParty::of(net)) {
        (true, Some(party)) => gpu_subtree(&party, inputs), // replaces :372-381
        _ => return d_acc_product_cpu(inputs, net, sid).await,
    };
    // gather of the roots (the forced 0, :381,:390) and the leader's top tree (:386-413): unchanged
    d_acc_product_leader(subtree, net, sid).await
}

pub async fn c_acc_product<F: FftField, Net: MPCSerializeNet>(
    inputs: &Vec<F>, pp: &PackedSharingParams<F>, net: &Net, sid: MultiplexedStreamID,
) -> Result<(Vec<F>, Option<Vec<F>>), MPCNetError> {
    let subtree = match (is_bls12_381_fr::<F>(), ZkParty::of(net)) {
        (true, Some(party)) => gpu_subtree(&party, inputs), // replaces :304-313
        _ => return c_acc_product_cpu(inputs, pp, net, sid).await,
    };
    // every party sends its last min(N_p, 2N) entries (:321-329), the leader interleaves and extends (:339-357): unchanged
    c_acc_product_leader(subtree, pp, net, sid).await
}
// `*_cpu`, `*_leader`: the reference's statements at the cited lines moved into helpers (not reproduced here).
