"""Host logic of the persistent depth encoder (ivln-ce_amd/depth_net.py), no GPU: the op table, the MFMA-order weight
packing and the statistics-partial wiring, replayed on the CPU by tests/depth_net_emulator.py (from the PACKED weights and
through the partial layout) against the oracle's torch restatement of habitat-lab's ResNetEncoder
(oracle/habitat_ext_ref.py; resnet_encoders.py:31-43, 95)."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.dirname(__file__))


def _encoder(seed=5):
    from ivln_ce_amd.encoders import ResNetEncoder

    torch.manual_seed(seed)
    enc = ResNetEncoder((256, 256, 1)).eval()
    with torch.no_grad():  # non-trivial affine parameters (the default init is gamma = 1, beta = 0)
        for m in enc.modules():
            if isinstance(m, torch.nn.GroupNorm):
                m.weight.normal_(1.0, 0.2)
                m.bias.normal_(0.0, 0.2)
    return enc


def test_weight_packing_round_trips_for_every_tile_form():
    import depth_net_emulator as E
    from ivln_ce_amd import depth_net as D

    g = torch.Generator().manual_seed(0)
    for (Cout, Cin, ks, M, KWT) in [(32, 1, 7, 16, 1), (32, 32, 1, 16, 4), (64, 64, 3, 16, 4), (256, 1024, 1, 8, 8), (256, 256, 3, 8, 8),
                                    (128, 1024, 3, 16, 32), (128, 32, 1, 16, 1), (512, 256, 1, 16, 2)]:
        w = torch.randn(Cout, Cin, ks, ks, generator=g)
        blob = D.pack_weights(w, M, KWT)
        assert torch.equal(E.unpack_weights(blob, Cout, Cin, ks, M, KWT), w), (Cout, Cin, ks, M, KWT)
        # the kernel's own index formula for one element: A[tile*M + i][kbeg + 4 chunk + u][kq]
        A = D.weight_matrix(w)
        per, cpk, ranges = D._k_ranges(A.shape[1], KWT)
        ent = 64 if M == 16 else 32
        tile, kwt, chunk, i, kq, u = Cout // M - 1, KWT - 1, 0, M - 1, 3, 1
        kb, ke = ranges[kwt]
        entry = i + (16 if M == 16 else 8) * kq
        got = blob[(((tile * KWT + kwt) * cpk + chunk) * ent + entry) * 4 + u]
        want = A[tile * M + i, kb + 4 * chunk + u, kq] if kb + 4 * chunk + u < ke else 0.0
        assert float(got) == float(want)


def test_program_covers_the_network_and_fits_the_cluster():
    import depth_net_emulator as E
    from ivln_ce_amd import depth_net as D

    prog = D.build_program(_encoder())
    convs = [o for o in prog.ops if o["kind"] == 0]
    assert len(convs) == 1 + 16 * 3 + 4 + 1 and prog.ops[-1]["kind"] == 1
    assert sum(o["barrier_before"] for o in prog.ops) == len(prog.ops) - 1 - 4  # every op but the stem and the 4 downsample convs
    for o in convs:
        assert o["WCT"] * o["WPT"] * o["KW"] == 8 and o["n_ctg"] * o["n_ptg"] * o["kwg"] <= 32
        assert o["n_ctg"] * o["WCT"] * o["M"] == o["Cout"]
        assert o["n_ptg"] * o["WPT"] * o["P"] * 16 == (1 << (2 * o["wout_shift"]))
        assert o["st_parts"] <= 32 and o["st_out_parts"] <= 32
        if o["ks"] != 7 and not (o["ks"] == 3 and o["stride"] == 2):
            assert o["cs"] % 32 == 16
    # algorithmic work: SURVEY 8d's 0.699 GFLOP per image for the depth encoder
    assert abs(prog.flops_per_image / 1e9 - 0.699) < 0.005, prog.flops_per_image
    assert prog.arena * 4 < 16 << 20  # per-image arena: every op owns its output (no reuse hazards); the live set is a few layers


def test_emulated_program_matches_the_oracle_encoder():
    import depth_net_emulator as E
    from ivln_ce_amd import depth_net as D
    from oracle import habitat_ext_ref as R

    enc = _encoder()
    prog = D.build_program(enc)
    space = types.SimpleNamespace(spaces={"depth": types.SimpleNamespace(shape=(256, 256, 1))})
    ref = R.ResNetEncoder(space, baseplanes=32, ngroups=16, make_backbone=R.resnet50)
    ref.load_state_dict(enc.state_dict())
    depth = torch.rand(1, 256, 256, 1, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want = ref({"depth": depth})[0]
        got = E.emulate(prog, depth[0, :, :, 0].contiguous())
    assert got.shape == want.shape == (128, 4, 4)
    err = float((got - want).abs().max())
    assert err < 2e-4, err


def test_supported_is_the_reference_default_architecture_only():
    """anything but the DD-PPO default (16 GroupNorm groups on 32 base planes, [3, 4, 6, 3] bottlenecks, a
    one-channel 7x7 stem) is declined up front - such an encoder runs the launch chain instead of tripping an assertion of
    `build_program` inside forward - and the verdict is cached on the encoder."""
    from ivln_ce_amd import depth_net as D
    from ivln_ce_amd.encoders import ResNetEncoder

    assert D.supported(_encoder())
    wide = ResNetEncoder((256, 256, 1), baseplanes=64, ngroups=32)
    assert not D.supported(wide) and not D.supported(wide)
    assert D.plan_for(wide, torch.device("cpu")) is None
    assert not D.supported(torch.nn.Linear(2, 2))
    prog = D.build_program(_encoder())
    assert len(prog.ops) == 55 and prog.arena * 4 < 16 << 20
