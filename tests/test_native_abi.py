"""
libzkhip.so from plain C (tests/native/abi_demo.c, gcc -std=c99 -pedantic): the header is a C header, the library a C-ABI
drop-in a compiled host can link -- the reference's Rust crates would bind the same symbols through rust/zkhip_sys.rs.
  * here (no GPU): it builds, links, loads, and FAILS LOUDLY (exit code 2) because there is no device and no CPU fallback;
  * on the GPU box: the program's self-consistency checks (linearity of the MSM, window table, batch / async batch, open == fold,
    zk_sumcheck_batch == single calls, error codes, zk_comm_init_all with one pthread per party) all pass.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


def _build():
    subprocess.check_call(["make", "-C", NATIVE, "-s"])
    return os.path.join(NATIVE, "abi_demo")


def test_c_program_builds_and_refuses_to_run_without_a_gpu():
    exe = _build()
    import zkhip

    if zkhip.lib().zk_device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_c_program_on_the_gpu():
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "abi_demo ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_latency_tool_builds_and_refuses_to_run_without_a_gpu():
    """tests/native/sc_latency.c (the C-ABI timing of the sumcheck family behind bench.py's `c_abi_2p20` leg)"""
    _build()
    exe = os.path.join(NATIVE, "sc_latency")
    assert os.path.exists(exe)
    import zkhip

    if zkhip.lib().zk_device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu test")
    r = subprocess.run([exe, "12"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "zk_ctx_create failed" in r.stderr, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_latency_tool_on_the_gpu():
    _build()
    r = subprocess.run([os.path.join(NATIVE, "sc_latency"), "12"], capture_output=True, text=True, timeout=300)
    lines = [l.split() for l in r.stdout.splitlines()]
    assert r.returncode == 0 and [l[0] for l in lines] == ["product", "plain", "fold", "open"], r.stdout + r.stderr
    assert all(l[2] == "mean" and float(l[3]) > 0 for l in lines)  # (the format bench.py parses)
