"""CPU emulation of a persistent-depth-encoder program (ivln_ce_amd.depth_net.build_program): test infrastructure, not a
fallback.  Replays the op table with torch on the CPU FROM THE PACKED WEIGHTS and through the statistics-partial layout, so
that the host-side packing and wiring of csrc/depth_net.hip are pinned without a GPU (tests/test_depth_net_program.py)."""
import torch
import torch.nn.functional as F

from ivln_ce_amd import depth_net as D


def unpack_weights(blob, Cout, Cin, ks, M, KWT):
    """Inverse of pack_weights (emulator / tests)."""
    ksteps = 13 if ks == 7 else (Cin // 4) * ks * ks
    per, cpk, ranges = D._k_ranges(ksteps, KWT)
    nct, ent = Cout // M, (64 if M == 16 else 32)
    v = blob.view(nct, KWT, cpk, 4, M, 4)  # tile, kwt, chunk, kq, i, u
    A = torch.zeros(Cout, ksteps, 4)
    for kwt, (kb, ke) in enumerate(ranges):
        n = ke - kb
        if n <= 0:
            continue
        blk = v[:, kwt].permute(0, 3, 1, 4, 2).reshape(Cout, cpk * 4, 4)  # tile, i, chunk, u, kq
        A[:, kb:ke] = blk[:, :n]
    if ks == 7:
        return A.reshape(Cout, 52)[:, :49].reshape(Cout, 1, 7, 7)
    KK = ks * ks
    return A.view(Cout, Cin // 4, KK, 4).permute(0, 1, 3, 2).reshape(Cout, Cin, ks, ks).contiguous()



def emulate(prog, depth):
    """depth (256, 256) float32 of ONE image -> (C, h, w) features, executing `prog` op by op with torch on the CPU."""
    arena = torch.zeros(prog.arena, dtype=torch.float32)
    wts = torch.cat(prog.wchunks)
    prm = torch.cat(prog.pchunks)
    eps = prog.eps
    out = None
    for op in prog.ops:
        if op["kind"] == 1:
            n = op["Cin"] * op["Hin"] * op["Win"]
            v = sum(arena[op["src_off"] + z * op["slab_stride"]: op["src_off"] + z * op["slab_stride"] + n] for z in range(op["nslab"]))
            v = v.view(1, op["Cin"], op["Hin"], op["Win"])
            out = F.relu(F.group_norm(v, 1, prm[op["gamma_off"]:op["gamma_off"] + op["Cin"]], prm[op["beta_off"]:op["beta_off"] + op["Cin"]], eps))[0]
            continue
        Cin, Cout, ks, s, pad, Hin, Win = (op[k] for k in ("Cin", "Cout", "ks", "stride", "pad", "Hin", "Win"))
        Wout = 1 << op["wout_shift"]

        def merged(st_off, parts, C_):
            st = arena[st_off: st_off + 16 * parts * 4].view(16, parts, 4)
            n, m, M2 = st[..., 0], st[..., 1], st[..., 2]
            cnt = n.sum(1)
            mean = (n * m).sum(1) / cnt
            var = (M2 + n * (m - mean[:, None]) ** 2).sum(1) / cnt
            return mean.repeat_interleave(C_ // 16), torch.rsqrt(var + eps).repeat_interleave(C_ // 16)

        if op["avg_in"]:
            x = F.avg_pool2d(depth.view(1, 1, 2 * Hin, 2 * Win), 2)
        else:
            Hr, Wr = (2 * Hin, 2 * Win) if op["pool"] else (Hin, Win)
            n = Cin * Hr * Wr
            x = sum(arena[op["src_off"] + z * op["slab_stride"]: op["src_off"] + z * op["slab_stride"] + n] for z in range(op["nslab"]))
            x = x.view(1, Cin, Hr, Wr).clone()
            if op["st_parts"]:
                mean, rstd = merged(op["st_off"], op["st_parts"], Cin)
                ga, be = prm[op["gamma_off"]:op["gamma_off"] + Cin], prm[op["beta_off"]:op["beta_off"] + Cin]
                x = (x - mean.view(1, -1, 1, 1)) * (rstd * ga).view(1, -1, 1, 1) + be.view(1, -1, 1, 1)
                if op["src2_off"] >= 0:
                    x2 = arena[op["src2_off"]: op["src2_off"] + n].view(1, Cin, Hr, Wr)
                    mean2, rstd2 = merged(op["st2_off"], op["st2_parts"], Cin)
                    g2, b2 = prm[op["gamma2_off"]:op["gamma2_off"] + Cin], prm[op["beta2_off"]:op["beta2_off"] + Cin]
                    x = x + (x2 - mean2.view(1, -1, 1, 1)) * (rstd2 * g2).view(1, -1, 1, 1) + b2.view(1, -1, 1, 1)
            if op["res_off"] >= 0:
                x = x + arena[op["res_off"]: op["res_off"] + n].view(1, Cin, Hr, Wr)
            if op["relu"]:
                x = F.relu(x)
            if op["pool"]:
                x = F.max_pool2d(x, 3, 2, 1)
            if op["act_out_off"] >= 0:
                arena[op["act_out_off"]: op["act_out_off"] + Cin * Hin * Win] = x.reshape(-1)
        KWT = op["KW"] * op["kwg"]
        ksteps = op["ksteps"]
        per, cpk, ranges = D._k_ranges(ksteps, KWT)
        ent = 64 if op["M"] == 16 else 32
        nblob = (Cout // op["M"]) * KWT * cpk * ent * 4
        w = unpack_weights(wts[op["w_off"]: op["w_off"] + nblob], Cout, Cin, ks, op["M"], KWT)
        HWo = Wout * Wout
        if op["kwg"] == 1:
            y = F.conv2d(x, w, None, s, pad)[0]
            arena[op["dst_off"]: op["dst_off"] + Cout * HWo] = y.reshape(-1)
        else:  # slabs: workgroup K slices = contiguous ranges of the packed k order
            A = D.weight_matrix(w)  # (Cout, ksteps, 4)
            for kg in range(op["kwg"]):
                kb, ke = ranges[kg * op["KW"]][0], ranges[(kg + 1) * op["KW"] - 1][1]
                Ak = torch.zeros_like(A)
                Ak[:, kb:ke] = A[:, kb:ke]
                KK = ks * ks
                wk = Ak.view(Cout, Cin // 4, KK, 4).permute(0, 1, 3, 2).reshape(Cout, Cin, ks, ks)
                arena[op["dst_off"] + kg * op["dst_slab_stride"]: op["dst_off"] + kg * op["dst_slab_stride"] + Cout * HWo] = \
                    F.conv2d(x, wk, None, s, pad)[0].reshape(-1)
            continue
        if op["st_out_parts"]:
            parts = op["st_out_parts"]
            st = arena[op["st_out_off"]: op["st_out_off"] + 16 * parts * 4].view(16, parts, 4)
            rows_t, PG, cpo = op["WCT"] * op["M"], 16 * op["WPT"] * op["P"], Cout // 16
            yf = y.reshape(Cout, HWo)
            for ctg in range(op["n_ctg"]):
                for ptg in range(op["n_ptg"]):
                    co0 = ctg * rows_t
                    rows_lg = min(cpo, rows_t)
                    cparts = max(1, cpo // rows_t)
                    part = ptg * cparts + ((co0 % cpo) // rows_t if cpo > rows_t else 0)
                    for lg in range(rows_t // rows_lg):
                        blk = yf[co0 + lg * rows_lg: co0 + (lg + 1) * rows_lg, ptg * PG:(ptg + 1) * PG]
                        g = co0 // cpo + (0 if cpo > rows_t else lg)
                        st[g, part, 0] = blk.numel()
                        st[g, part, 1] = blk.mean()
                        st[g, part, 2] = ((blk - blk.mean()) ** 2).sum()
    return out


