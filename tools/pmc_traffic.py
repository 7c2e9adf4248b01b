"""HBM traffic per env step / per MFMA-family launch from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE;
tools/pmc_stats.py CSVs):  python tools/pmc_traffic.py <fetch.csv> <write.csv> <steps> <out.json> [flags] [steps_mfma]
Counter values are KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies wide coalesced
reads at half their bytes), WRITE_SIZE taken as reported.

The family is tools/kernel_family.py's list (= `ivln_family_kernel_names()` of the library = the launches bench.py's
FLOP hooks and duration sink see): round 5's prefix list missed k_conv1x1_bf3_ks and under-reported the step by 1.75 GB.
`by_kernel` carries every member's launches / bytes per step so the figure can be recomputed from the json alone, and
tests/test_gpu_bench.py compares its launch counts with the live hook's."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_family import FAMILY_KERNELS, MAPPER_KERNELS, base_name  # noqa: E402


def load(path):
    """{kernel base name: [launches, KiB]} and the pass total in bytes."""
    per, tot = {}, 0.0
    for r in csv.DictReader(open(path)):
        a = per.setdefault(base_name(r["Name"]), [0, 0.0])
        a[0] += int(r["Launches"])
        a[1] += float(r["Total"])
        tot += float(r["Total"])
    return per, tot * 1024


def family_sum(per, members):
    fam = {k: v for k, v in per.items() if k in members}
    return sum(v[1] for v in fam.values()) * 1024, sum(v[0] for v in fam.values())


def main():
    fetch, write, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    flags = sys.argv[5] if len(sys.argv) > 5 else "--envs 4"
    # steps in which the MFMA family ran at all (bench.py's mapper_roofline passes launch the mapper only)
    steps_mfma = int(sys.argv[6]) if len(sys.argv) > 6 else steps
    fper, ft = load(fetch)
    wper, wt = load(write)
    fm, fl = family_sum(fper, FAMILY_KERNELS)
    wm, _ = family_sum(wper, FAMILY_KERNELS)
    mpf, mpl = family_sum(fper, MAPPER_KERNELS)
    mpw, _ = family_sum(wper, MAPPER_KERNELS)
    lps = fl / steps_mfma
    by_kernel = {}
    for k in FAMILY_KERNELS:
        if k in fper or k in wper:
            f = fper.get(k, [0, 0.0])
            w = wper.get(k, [0, 0.0])
            by_kernel[k] = {
                "launches_per_step": round(f[0] / steps_mfma, 3),
                "hbm_bytes_per_step_corrected": int((2 * f[1] + w[1]) * 1024 / steps_mfma),
            }
    others = sorted(((k, v) for k, v in fper.items() if k not in FAMILY_KERNELS),
                    key=lambda kv: -(2 * kv[1][1] + wper.get(kv[0], [0, 0.0])[1]))[:8]
    d = {
        "workload": f"bench.py {flags} --no-graph (eager launches; PMC serialises kernels), {steps} steps",
        "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py {flags} --no-cpu-baseline "
                   "--no-update --no-collect --no-gt-leg --no-graph (and a second pass with --pmc WRITE_SIZE)",
        "correction": "gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section: wide coalesced reads are tallied at "
                      "half their bytes); WRITE_SIZE as reported (uncalibrated)",
        "family_definition": "tools/kernel_family.py FAMILY_KERNELS == ivln_family_kernel_names(): " + ", ".join(FAMILY_KERNELS),
        "mfma_family": {
            "launches_per_step": round(lps, 2),
            "steps": steps_mfma,
            "fetch_bytes_per_step_raw": int(fm / steps_mfma),
            "write_bytes_per_step": int(wm / steps_mfma),
            "hbm_bytes_per_step_corrected": int((2 * fm + wm) / steps_mfma),
            "hbm_bytes_per_launch_corrected": int((2 * fm + wm) / steps_mfma / lps) if lps else None,
            "by_kernel": by_kernel,
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
        "largest_non_family_kernels_bytes_per_step": {
            k: int((2 * v[1] + wper.get(k, [0, 0.0])[1]) * 1024 / steps) for k, v in others},
    }
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d["mfma_family"]))


if __name__ == "__main__":
    main()
