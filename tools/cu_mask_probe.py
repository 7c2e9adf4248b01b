"""gt-semantics step with the main graphs (mapper / map CNN / instruction, then the head) replayed on a CU-MASKED stream, the
persistent depth encoder on its own unmasked stream: at 4 envs the encoder's clusters sit on four XCDs - do the other graphs
run better on the CUs of the XCDs it leaves free?  Masks are given as a byte pattern repeated over the 256 CU bits.
python tools/cu_mask_probe.py [envs]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import gen_observations, make_policy  # noqa: E402
from ivln_ce_amd.graphed import GraphedRollout  # noqa: E402
from ivln_ce_amd.obs_transforms import GTSemanticsIterativeMapper  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
cfg, policy = make_policy(dev)
tr = GTSemanticsIterativeMapper.from_config(cfg)
obs = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in o.items()} for o in gen_observations(B, 40, 1)]
r = GraphedRollout(policy, [tr], obs[0], deterministic=True, streams="split")
hip = C.CDLL("libamdhip64.so")


def masked_stream(pattern):
    if pattern is None:
        return torch.cuda.Stream()
    words = (C.c_uint32 * 8)(*([int.from_bytes(bytes([pattern] * 4), "little")] * 8))
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def run(stream, n=200):
    with torch.cuda.stream(stream):
        for i in range(20):
            r.step(obs[i % 40])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            r.step(obs[i % 40])
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / n


print(f"envs {B}; us per step")
for name, pat in (("default stream", "cur"), ("a plain second stream", None), ("mask 0xFF (all CUs)", 0xFF), ("mask 0xF0", 0xF0), ("mask 0x0F", 0x0F),
                  ("mask 0x78", 0x78), ("mask 0x87", 0x87), ("mask 0xAA", 0xAA), ("default stream again", "cur")):
    st = torch.cuda.current_stream() if pat == "cur" else masked_stream(pat)
    print(f"  {name:<26} {run(st):7.1f}", flush=True)
