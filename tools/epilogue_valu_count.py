"""Round-3 verdict item 3 (i): where do the instructions of ffn.0's epilogue (bias + GELU-tanh + 128x128-block quantiser, the QOUT
instantiation of gemm_w8a8_fi_kernel) go?  Static count from the compiler's own assembly: an analysis build with assembler
comments at the phase boundaries (-DTD_PHASE_MARKS; the comments emit no instruction), instructions between the marks counted
by class.  Per lane the epilogue handles 128 output elements (wave tile 128 x 64 over 64 lanes), so counts / 128 = per element.

    python tools/epilogue_valu_count.py        (needs hipcc; no GPU)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "turbodiffusion_amd", "csrc", "gemm_w8a8_fi.hip")
KERN = "_Z19gemm_w8a8_fi_kernelILi1ELi1ELb1ELi0ELi0ELb1ELb0ELi0ELb0ELi0EEv"      # <BF16, GELU_TANH, bias, 0, 0, QOUT>


def klass(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")):
        return "valu_transcendental"
    if op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane", "ds_bpermute", "ds_swizzle")) or "_dpp" in op:
        return "cross_lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
        return "wait_barrier_nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
               "-DTD_PHASE_MARKS", "--cuda-device-only", "-S", SRC, "-o", out]
        subprocess.run(cmd, check=True)
        text = open(out).read()
    m = re.search(r"^" + re.escape(KERN) + r".*?:\n(.*?)\n\s*s_endpgm", text, re.S | re.M)
    if not m:
        sys.exit("kernel not found in the assembly")
    phase, counts = "main_loop_and_before", collections.OrderedDict()
    for line in m.group(1).splitlines():
        ls = line.strip()
        pm = re.match(r"; TD_PHASE (\w+)", ls)
        if pm:
            phase = pm.group(1)
            continue
        if not ls or ls.startswith((";", ".", "//")) or ls.endswith(":"):
            continue
        op = ls.split()[0]
        counts.setdefault(phase, collections.Counter())[klass(op)] += 1
    names = {"qout_begin": "(1) cast + bias + GELU-tanh -> 16-bit, 128 elements per lane", "qout_amax": "(2) amax of the wave's half block",
             "qout_exchange": "(3) exchange with the partner wave, scale, multiplier", "qout_quantise_store": "(4) quantise, lane swaps, 16-byte stores",
             "qout_end": "(after the epilogue)"}
    print(f"{'phase':66s} {'valu':>6s} {'trans':>6s} {'xlane':>6s} {'lds':>5s} {'vmem':>5s} {'salu':>5s}   VALU+trans per element (/128)")
    tot = collections.Counter()
    for ph, c in counts.items():
        if not ph.startswith("qout") or ph == "qout_end":
            continue
        tot.update(c)
        v = c["valu"] + c["valu_transcendental"] + c["cross_lane"]
        print(f"{names.get(ph, ph):66s} {c['valu']:6d} {c['valu_transcendental']:6d} {c['cross_lane']:6d} {c['lds']:5d} {c['vmem']:5d} {c['salu']:5d}   {v / 128:6.2f}")
    v = tot["valu"] + tot["valu_transcendental"] + tot["cross_lane"]
    print(f"{'QOUT epilogue, total':66s} {tot['valu']:6d} {tot['valu_transcendental']:6d} {tot['cross_lane']:6d} {tot['lds']:5d} {tot['vmem']:5d} {tot['salu']:5d}   {v / 128:6.2f}")
    ml = counts.get("main_loop_and_before", collections.Counter())
    print(f"(main loop + prologue, static: {ml['mfma']} MFMA, {ml['valu']} VALU, {ml['lds']} LDS, {ml['vmem']} VMEM instructions in the code, the K loop runs 12 x for ffn.0)")


if __name__ == "__main__":
    main()
