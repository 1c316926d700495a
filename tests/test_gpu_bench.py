"""
bench.py launched EXACTLY as the driver launches it -- `python bench.py --gpus N ...`, no launcher -- on the GPU box.

The box has one GPU, so N > 1 runs are functional runs (ZK_BENCH_BACKEND = gloo / local: ranks share GPU 0, exchanges
staged through the host): they prove that the N > 1 code path starts, shards, exchanges, checks itself and prints its
line; the only thing an 8-GPU node adds is the wire (RCCL instead of gloo / the thread net).  With the default backend
and fewer GPUs than ranks the line must carry "error".
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None, timeout=900):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update({"HSA_ENABLE_IPC_MODE_LEGACY": "0", "ZK_BENCH_DEADLINE_S": "600"})
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=e)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_more_ranks_than_gpus_is_an_error_line():
    import zkhip  # (not torch: importing it here would swap the HIP runtime / RCCL copy under every later test of the session)

    found = zkhip.lib().zk_device_count()
    n = 2 * max(found, 1)
    r, line = _bench("--gpus", str(n), "--no-cpu")
    assert r.returncode != 0 and "Traceback" not in r.stderr, r.stderr[-2000:]
    assert line["error"] == f"needs {n} GPUs, found {found}" and line["n_gpus"] == n


def test_gpus_2_self_launches_processes_over_gloo():
    """`python bench.py --gpus 2` re-executes itself under torch.distributed.run; one MSM over 2 x 2^16 points in two chunks"""
    r, line = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu", "--log2n", "16", "--big", "18", env={"ZK_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert [d["rank"] for d in line["ranks"]["devices"]] == [0, 1] and len({d["pid"] for d in line["ranks"]["devices"]}) == 2
    assert "torch.distributed gloo" in line["exchange_backend"] and line["rccl_ranks"] == 0
    assert "legs_error" not in line and line["strong"]["msm_2p18"]["points_per_rank"] == 1 << 17
    assert line["strong"]["sumcheck_product_2p18"]["layout"].startswith("cyclic")


def test_gpus_8_party_threads_with_anchored_e2e():
    """one process, eight party threads (--party-threads): the 8-party d_msm as the step, and the 8-party protocol (n = 12) with
    every transcript chain anchored by all parties' values -- the check the 8-GPU run performs"""
    r, line = _bench("--gpus", "8", "--party-threads", "--steps", "2", "--warmup", "1", "--no-cpu", "--log2n", "12", "--big", "14", "--e2e-n", "12",
                     env={"ZK_BENCH_BACKEND": "local"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert line["n_gpus"] == 8 and line["value"] > 0 and line["config"]["parties"] == 8
    assert line["ranks"]["model"].startswith("one process, one host thread per party") and len(line["ranks"]["devices"]) == 8
    assert "legs_error" not in line, line.get("legs_error")
    e2e = line["e2e"]
    assert "error" not in e2e, e2e
    assert e2e["transcript_checks"] == "ok" and e2e["transcript_check_kind"].startswith("anchored") and "all 8 parties" in e2e["transcript_check_kind"]
    assert e2e["timers_s"]["Distributed HyperPlonk"] > 0 and e2e["timers_s_serial_steps"]["Distributed HyperPlonk"] > 0


def test_gpus_8_line_carries_the_compiled_hosts_rccl_mode_proof():
    """the N = 8 line's e2e_cpp_rccl leg: rank 0 starts `hyperplonk --mode rccl --check` (ONE process for all 8 parties) while the bench's
    own ranks wait; here over the test double of librccl (--share-gpus), on the node over the real one.  Every party's verdict is in
    the line and the digest equals the Python host's 8-party run"""
    fake = os.path.join(ROOT, "tests", "native", "fake_rccl")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "native"), "-s", "fake_rccl/librccl.so.1"])
    r, line = _bench("--gpus", "8", "--party-threads", "--steps", "2", "--warmup", "1", "--no-cpu", "--log2n", "12", "--big", "14", "--e2e-n", "12",
                     env={"ZK_BENCH_BACKEND": "local", "ZK_BENCH_CPP_RCCL": "share", "LD_LIBRARY_PATH": fake + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    leg = line["e2e_cpp_rccl"]
    assert "error" not in leg and set(leg) == {"n12", "note"}, leg
    c = leg["n12"]
    assert "error" not in c, c
    assert c["self_check_ok"] is True and len(c["self_check"]) == 8 and all(" ok -- anchored" in v for v in c["self_check"])
    assert c["transcript_equals_python_host"] is True and c["transcript_sha256"] == [line["e2e"]["transcript_sha256"]]
    assert 0 < c["timers_s"]["Distributed HyperPlonk"] < 5 and "fake_rccl" in c["stderr_tail"]


def test_single_gpu_line_has_the_record_fields():
    r, line = _bench("--steps", "3", "--warmup", "1", "--no-cpu", "--big", "22", "--e2e-n", "12", timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 0 and line["exchange_backend"].startswith("none")
    rf = line["roofline"]
    assert 0 < rf["frac_of_architectural"] < rf["frac"] < 1 and abs(rf["peak_architectural"] - 39.3) < 0.1
    assert 0 < rf["hbm"]["frac"] < 1 and rf["hbm"]["algorithmic_bytes_per_launch"] == 128.0 * (1 << 20)
    ds = line["d_sumcheck"]
    assert ds["unit"] == "Fr field-ops/s" and ds["value"] > 1e9 and set(ds["roofline"]) == {"product", "plain"}
    assert 0 < ds["roofline"]["product"]["hbm"]["frac"] < 1 and 0 < ds["roofline"]["product"]["int_alu"]["frac"] < 1
    r24 = line["roofline_msm_2p22"]
    assert r24["points"] == 1 << 22 and 0 < r24["frac"] < 1 and r24["kernel_ms"] > 0
    e2e = line["e2e"]
    assert e2e["transcript_checks"] == "ok" and e2e["transcript_check_kind"].startswith("anchored")
    assert e2e["scalar_muls_computed"] == 97227 - (1 << 13)
    # the same proof from the compiled C++ host, self-checked in the run: its transcript digest equals the Python driver's
    cpp = e2e["cpp_host"]
    assert "error" not in cpp and cpp["transcript_equals_python_host"] is True and len(cpp["transcript_sha256"]) == 1
    assert e2e["proof_s"]["host"] in ("python", "cpp") and e2e["proof_s"]["seconds"] == min(e2e["proof_s"]["all"].values())
    assert 0 < cpp["timers_s"]["Distributed HyperPlonk"] < 1 and cpp["comm_per_proof"] == "(959224, 959224)"
    # the collaborative permutation check alone, compiled host, self-checked (5 commits / opens recomputed by single calls)
    cp = line["cpermcheck"]
    assert "error" not in cp and cp["self_check_ok"] is True and "5 recomputed" in cp["self_check"], cp
    assert 0 < cp["timers_s"]["Collaborative Permcheck"] < 1 and cp["scalar_muls_computed"] == 20 * (4 << 12) - 10
