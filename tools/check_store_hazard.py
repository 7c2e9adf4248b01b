"""Static check behind BF3_STORE_GUARD (csrc/conv_bf3.hip, tools/store_hazard.hip): every buffer / global store of more than 64
bits in the built gfx950 code whose soffset is an SGPR - the form the compiler's hazard recognizer does NOT pad - must be
followed by enough wait states before any VALU / VMEM-load write of its data registers.

    python tools/check_store_hazard.py [--need N] [objects ...]      (default: ivln-ce_amd/csrc/*.o, need = 2 wait states)

Walks the disassembly of each object's gfx950 code object; for each wide store with a register soffset it counts wait
states (an instruction = 1, s_nop k = k + 1) until the first later instruction that writes one of the store's data
registers, staying inside the straight-line run (a branch / label / s_endpgm ends the window: conservative - the window is
then reported as 'leaves the block').  Prints every site with fewer than N wait states; exit status 1 if there is any."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STORE = re.compile(r"^(buffer_store_dwordx[34]|buffer_store_b(96|128))\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")
VDST = re.compile(r"^\s*(v_\w+|buffer_load\w*|global_load\w*|ds_read\w*|ds_load\w*|flat_load\w*|scratch_load\w*)\s+(v\d+|v\[\d+:\d+\]|a\d+|a\[\d+:\d+\])")


def disassemble(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, dev = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        if obj.endswith(".co"):
            dev = obj
        else:
            subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(td, "unused.o")])
            subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={dev}"])
        return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", dev], text=True)


def regs_of(tok):
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"[va](\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(text, need):
    bad, sites = [], 0
    kernel = "?"
    lines = text.splitlines()
    ins = []  # (kernel, text)
    for ln in lines:
        m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
        if m:
            kernel = m.group(1)
            ins.append((kernel, "<label>"))
            continue
        t = ln.strip()
        if not t or t.startswith(("Disassembly", "/")) or ":" in t.split()[0] and not t.split()[0].startswith(("v_", "s_", "buffer", "global", "ds_", "flat", "scratch")):
            continue
        t = t.split("//")[0].strip()
        if t:
            ins.append((kernel, t))
    for i, (k, t) in enumerate(ins):
        m = STORE.match(t)
        if not m:
            continue
        soff = m.group(6).rstrip(",")
        if not re.match(r"^s\d+$", soff):  # literal / 'off' / inline constant: the compiler pads these itself
            continue
        sites += 1
        data = set(range(int(m.group(3)), int(m.group(4)) + 1))
        waits, verdict = 0, None
        for k2, t2 in ins[i + 1:i + 1 + 12]:
            if t2 == "<label>" or t2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                verdict = "leaves the block" if waits < need else None
                break
            d = VDST.match(t2)
            if d and d.group(2).startswith("v") and regs_of(d.group(2)) & data and d.group(1).startswith("v_"):
                verdict = f"VALU write of {d.group(2)} after {waits} wait state(s): {t2}" if waits < need else None
                break
            if waits >= need:
                break
            nop = re.match(r"^s_nop\s+(\d+)", t2)
            waits += (int(nop.group(1)) + 1) if nop else 1
        if verdict:
            bad.append((k, t, verdict))
    return sites, bad


def main():
    args = sys.argv[1:]
    need = 2
    if "--need" in args:
        i = args.index("--need")
        need = int(args[i + 1])
        del args[i:i + 2]
    objs = args or sorted(glob.glob(os.path.join(ROOT, "ivln-ce_amd", "csrc", "*.o")))
    total_bad = 0
    for o in objs:
        try:
            text = disassemble(o)
        except subprocess.CalledProcessError:
            print(f"{os.path.basename(o)}: no gfx950 code object")
            continue
        sites, bad = check(text, need)
        print(f"{os.path.basename(o)}: {sites} wide store(s) with a register soffset, {len(bad)} with fewer than {need} wait states behind them")
        for k, t, v in bad:
            print(f"    {k[:70]}: {t}\n        -> {v}")
        total_bad += len(bad)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
