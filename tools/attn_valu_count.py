"""Round-4 verdict item 5: where do the ~9 VALU-class instructions per MFMA of the sparse attention kernel go?  Static count from
the compiler's own assembly of the production instantiation (INT8 QK^T, FP16 PV, three workgroups per CU): the K-tile loop is
located as the innermost backward branch around the MFMAs and its instructions are counted by opcode class — per wave and
64-key tile (24 MFMAs: 8 x v_mfma_i32_32x32x32_i8 for S^T = K Q^T, 16 x v_mfma_f32_32x32x16_f16 for O^T += V^T P^T; every lane
owns 32 scores of the tile).

    python tools/attn_valu_count.py        (needs hipcc; no GPU)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "turbodiffusion_amd", "csrc", "attn.hip")
# attn_kernel<QK_I8 = true, PDT = f16, ODT = bf16, ...>: the production sparse kernel (first instantiation that matches)
PAT = r"^(_Z11attn_kernelILb1ELi0ELi1ELb0ELb0ELb0ELb0E\w*):[^\n]*\n(.*?)\n\s*s_endpgm"

CLASSES = [
    ("mfma", lambda o: o.startswith("v_mfma")),
    ("exp2 (transcendental)", lambda o: o.startswith("v_exp")),
    ("other transcendental (rcp, ...)", lambda o: o.startswith(("v_rcp", "v_log", "v_sqrt", "v_rsq"))),
    ("max (v_max_f32 / v_max3_f32 / v_pk_max)", lambda o: o.startswith(("v_max", "v_pk_max"))),
    ("fma / mad (scale + offset of the exponent argument, rescales)", lambda o: o.startswith(("v_fma", "v_fmac", "v_mad", "v_pk_fma"))),
    ("add / sub f32 (row sums)", lambda o: o.startswith(("v_add_f32", "v_sub_f32", "v_pk_add_f32", "v_subrev_f32"))),
    ("mul f32 (accumulator rescale, scales)", lambda o: o.startswith(("v_mul_f32", "v_pk_mul_f32"))),
    ("convert / pack to fp16 (P operand)", lambda o: o.startswith(("v_cvt", "v_pack", "v_perm"))),
    ("cross-lane (permlane, readlane, dpp, bpermute)", lambda o: o.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane", "ds_bpermute", "ds_swizzle")) or "_dpp" in o),
    ("compare / select / mov", lambda o: o.startswith(("v_cmp", "v_cndmask", "v_mov", "v_accvgpr"))),
    ("integer / address VALU", lambda o: o.startswith("v_")),
    ("LDS reads (K / V^T fragments)", lambda o: o.startswith("ds_read") or o.startswith("ds_load")),
    ("LDS other", lambda o: o.startswith("ds_")),
    ("VMEM (LDS-DMA pieces, LUT)", lambda o: o.startswith(("global_", "buffer_", "flat_", "scratch_"))),
    ("waits / barriers / nops", lambda o: o.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep"))),
    ("scalar", lambda o: o.startswith("s_")),
]


def klass(op):
    for name, f in CLASSES:
        if f(op):
            return name
    return "other"


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
               "--cuda-device-only", "-S", SRC, "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    m = re.search(PAT, text, re.S | re.M)
    if not m:
        sys.exit("kernel not found in the assembly")
    name, body = m.group(1), m.group(2).splitlines()
    # labels and backward branches
    pos = {}
    for i, ln in enumerate(body):
        lm = re.match(r"^(\.LBB\d+_\d+):", ln.strip())
        if lm:
            pos[lm.group(1)] = i
    loops = []
    for i, ln in enumerate(body):
        bm = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln)
        if bm and bm.group(1) in pos and pos[bm.group(1)] < i:
            blk = body[pos[bm.group(1)]:i + 1]
            nm = sum(1 for b in blk if b.strip().startswith("v_mfma"))
            if nm:
                loops.append((len(blk), nm, pos[bm.group(1)], i))
    if not loops:
        sys.exit("no loop with MFMAs found")
    # the K-tile loop: the SMALLEST backward-branch region that holds all 24 MFMAs of a tile
    cands = [l for l in loops if l[1] >= 24]
    n, nm, a, b = min(cands) if cands else max(loops, key=lambda l: l[1])
    # basic blocks of the loop; a forward branch over >= 50 instructions marks a RARE path (the tail mask of the last, partial K
    # block; the accumulator rescale, taken only when a row maximum grows by more than 2^8) — counted apart
    lines = [l.strip() for l in body[a:b + 1]]
    lab = {}
    for i, ls in enumerate(lines):
        lm = re.match(r"^(\.LBB\d+_\d+):", ls)
        if lm:
            lab[lm.group(1)] = i
    rare = [False] * len(lines)
    for i, ls in enumerate(lines):
        bm = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ls)
        if bm and bm.group(1) in lab and lab[bm.group(1)] > i:
            span = [j for j in range(i + 1, lab[bm.group(1)]) if lines[j] and not lines[j].startswith((";", ".", "//")) and not lines[j].endswith(":")]
            if len(span) >= 50 and not any(lines[j].startswith(("buffer_", "global_")) for j in span):   # (the tile fetch is common)
                for j in span:
                    rare[j] = True
    cnt, cnt_rare = collections.Counter(), collections.Counter()
    for i, ls in enumerate(lines):
        if not ls or ls.startswith((";", ".", "//")) or ls.endswith(":"):
            continue
        (cnt_rare if rare[i] else cnt)[klass(ls.split()[0])] += 1
    mf = cnt["mfma"]
    valu_names = [c for c, _ in CLASSES[1:11]]
    valu = sum(cnt[c] for c in valu_names)
    valu_r = sum(cnt_rare[c] for c in valu_names)
    print(f"kernel {name[:60]}...: K-tile loop = {sum(cnt.values()) + sum(cnt_rare.values())} instructions, {mf} MFMAs per wave and tile")
    print(f"{'class':64s} {'common':>7s} {'per MFMA':>9s} {'per score (/32)':>16s} {'rare paths':>11s}")
    for c, _ in CLASSES:
        if cnt[c] or cnt_rare[c]:
            print(f"{c:64s} {cnt[c]:7d} {cnt[c] / mf:9.2f} {cnt[c] / 32:16.2f} {cnt_rare[c]:11d}")
    print(f"{'VALU-class (everything v_* except MFMA), total':64s} {valu:7d} {valu / mf:9.2f} {valu / 32:16.2f} {valu_r:11d}")
    print(f"common path: {valu + mf} VALU-class + MFMA issue slots per tile = {(valu + mf) / mf:.2f} per MFMA; matrix work {mf} x 32 = {mf * 32} cycles; "
          f"the transcendentals alone: {cnt['exp2 (transcendental)']} x 16 = {cnt['exp2 (transcendental)'] * 16} cycles of the SIMD's VALU pipe per wave and tile")
    print("rare paths: the tail mask (last, partial K block only) and the accumulator rescale (a row maximum grew by more than 2^8)")


if __name__ == "__main__":
    main()
