import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The host sides of the end-to-end tests (synthetic env, batch_obs) are thousands of small CPU tensor ops; with one
    # OpenMP thread per core of a host shared with other jobs each of them pays a crowded barrier (torch.stack of two
    # depth frames: 1.8 ms instead of 0.1 - 400 s instead of 120 for the GPU suite on a busy box).  The oracle ports that
    # want more threads set them themselves.
    try:
        import torch

        torch.set_num_threads(min(4, max(1, torch.get_num_threads())))
    except Exception:  # noqa: BLE001
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (*.so are git-ignored): build the HIP library and the C oracle
    once (hipcc cross-compiles gfx950 without a GPU; ~1-2 minutes the first time, a no-op afterwards)."""
    need = [os.path.join(ROOT, "ivln-ce_amd", "libivln_hip.so"), os.path.join(ROOT, "oracle", "libmapper_ref.so")]
    if all(os.path.exists(p) for p in need):
        return
    import __graft_entry__ as ge

    ge.build()


@pytest.fixture
def same_depth_path():
    """Tests that demand BIT-identical results from two executions of the depth encoder pin it to ONE implementation for
    both: the persistent launch (csrc/depth_net.hip, the default up to 8 images where the encoder is latency-bound) and
    the launch chain are the same arithmetic in a different summation order (1e-5 apart), and which of them runs depends
    on batch size, IVLN_DEPTH_NET_SPLIT_MIN and whether semantics are predicted.  Yields a function that sets the mode;
    restored afterwards."""
    from ivln_ce_amd import ops

    old = ops.DEPTH_NET

    def pin(mode):
        ops.DEPTH_NET = mode

    yield pin
    ops.DEPTH_NET = old


_RENDEZVOUS_NOISE = ("address already in use", "EADDRINUSE", "RendezvousConnectionError", "DistNetworkError", "Connection reset",
                     "Connection refused", "Socket Timeout", "connect() timed out", "Broken pipe")


def free_port():
    """A TCP port that is free on 127.0.0.1 right now."""
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def run_torchrun(nproc, script, args=(), env=None, timeout=900):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node nproc script args` on a port that is free NOW (a fixed port
    collides with a socket another process on the box still holds - seen once in ~40 runs of the two-rank tests), started
    once more, on another port, when the ranks died of the rendezvous itself; a failure of the test's own assertions is
    returned as it is.  -> subprocess.CompletedProcess"""
    import subprocess

    r = None
    for attempt in range(2):
        port = free_port()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), script, *[str(a) for a in args]]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        if r.returncode == 0 or not any(m in (r.stderr + r.stdout) for m in _RENDEZVOUS_NOISE):
            break
    return r
