"""HBM traffic per env step / per MFMA-family launch from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE;
tools/pmc_stats.py CSVs):  python tools/pmc_traffic.py <fetch.csv> <write.csv> <steps> <out.json>
Counter values are KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies wide coalesced
reads at half their bytes), WRITE_SIZE taken as reported."""
import csv
import json
import sys

MFMA = ("k_gemm", "k_conv_direct", "k_conv_bf3", "k_conv1x1_stream", "k_wgrad", "k_conv_gn", "k_gn_conv", "k_nconv", "k_depth_net")
MAPPER = ("k_local_", "k_world_", "k_finalize", "k_frames")


def load(path, family=MFMA):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["Total"]) for r in rows) * 1024
    fam = [r for r in rows if r["Name"].startswith(family) and "_pack" not in r["Name"]]  # (weight-packing helpers are not MFMA launches)
    mf = sum(float(r["Total"]) for r in fam) * 1024
    launches = sum(int(r["Launches"]) for r in fam)
    return tot, mf, launches


fetch, write, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
flags = sys.argv[5] if len(sys.argv) > 5 else "--envs 4"
# steps in which the MFMA family ran at all (bench.py's mapper_roofline passes launch the mapper only): round 4 divides the
# family's bytes by THIS count - rounds 1-3 divided by all traced steps, which understated the gt-semantics step's family
# traffic by 65 / 45 (the committed r01-r03 figures are kept as they were reported)
steps_mfma = int(sys.argv[6]) if len(sys.argv) > 6 else steps
ft, fm, fl = load(fetch)
wt, wm, _ = load(write)
lps = fl / steps_mfma
_, mpf, mpl = load(fetch, MAPPER)
_, mpw, _ = load(write, MAPPER)
d = {
    "workload": f"bench.py {flags} --no-graph (eager launches; PMC serialises kernels), {steps} steps",
    "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py {flags} --no-cpu-baseline "
               "--no-update --no-pred-leg --no-graph (and a second pass with --pmc WRITE_SIZE)",
    "correction": "gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section: wide coalesced reads are tallied at "
                  "half their bytes); WRITE_SIZE as reported (uncalibrated)",
    "mfma_family": {
        "launches_per_step": round(lps, 2),
        "steps": steps_mfma,
        "fetch_bytes_per_step_raw": int(fm / steps_mfma),
        "write_bytes_per_step": int(wm / steps_mfma),
        "hbm_bytes_per_step_corrected": int((2 * fm + wm) / steps_mfma),
        "hbm_bytes_per_launch_corrected": int((2 * fm + wm) / steps_mfma / lps),
    },
    "mapper": {
        "steps": steps,
        "launches_per_step": round(mpl / steps, 2),
        "fetch_bytes_per_step_raw": int(mpf / steps),
        "write_bytes_per_step": int(mpw / steps),
        "hbm_bytes_per_step_corrected": int((2 * mpf + mpw) / steps),
    },
    "all_kernels": {
        "fetch_bytes_per_step_raw": int(ft / steps),
        "write_bytes_per_step": int(wt / steps),
        "hbm_bytes_per_step_corrected": int((2 * ft + wt) / steps),
    },
    "note": "write traffic of the MFMA family is dominated by partial slabs (deferred split-K convs; the 16 per-group "
            "slabs of every k_gn_conv launch, reduced inside the consuming GroupNorm kernel): latency at 4 envs is bought "
            "with slab bytes",
}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d["mfma_family"]))
