"""Write ablated copies of csrc/gemm_w8a8_fi.hip to /tmp/abl/ (one ingredient of the main loop compiled out by replacing a
macro definition; results are garbage, only time and energy are read) — input of tools/build_variant.sh and
tools/gpu/gemm_ablation.sh:

    python tools/gemm_ablate_sources.py
    for v in nodeq nomfma nolds nodeq_nolds nomfma_nodeq; do
        bash tools/build_variant.sh gemm_w8a8_fi.hip /tmp/abl/gemm_$v.hip turbodiffusion_amd/libtd_abl_$v.so; done
    gpurun -- bash tools/gpu/gemm_ablation.sh        ->  profiles/r03_gemm_energy_ablation.txt"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(HERE, "..", "turbodiffusion_amd", "csrc", "gemm_w8a8_fi.hip")).read()
os.makedirs("/tmp/abl", exist_ok=True)


def macro(name):
    """the full text of `#define name(...)` including its continuation lines"""
    a = src.index("#define " + name + "(")
    b = a
    while True:
        e = src.index("\n", b)
        if src[e - 1] != "\\":
            return src[a:e]
        b = e + 1


NO_DEQ = [(macro("F_ADD4"), '#define F_ADD4(v_) asm volatile("" : "+v"((v_)[0]));'),
          (macro("F_FMAC4"), '#define F_FMAC4(acc_, v_, sc_) asm volatile("" : "+v"((acc_)[0]) : "v"((v_)[0]), "s"(sc_));')]
NO_MFMA = [(macro("F_MFMA0"), '#define F_MFMA0(d_, a_, b_) asm volatile("" : "=&v"(d_) : "v"(a_), "v"(b_), "v"(magic));'),
           (macro("F_MFMA1"), '#define F_MFMA1(d_, a_, b_) asm volatile("" : "+v"(d_) : "v"(a_), "v"(b_));')]
NO_LDS = [(macro("F_LOAD_X"), '#define F_LOAD_X(st_, i_, slot_) _Pragma("unroll") for (int kc = 0; kc < 2; ++kc) '
                              'asm volatile("" : "=v"(xf[slot_][kc]) : "v"(xoff[kc]));'),
          (macro("F_LOAD_W"), '#define F_LOAD_W(st_, kc_) _Pragma("unroll") for (int j = 0; j < 4; ++j) '
                              'asm volatile("" : "=v"(wf[j][kc_]) : "v"(woff[kc_]));')]
for name, edits in (("nodeq", NO_DEQ), ("nomfma", NO_MFMA), ("nolds", NO_LDS), ("nodeq_nolds", NO_DEQ + NO_LDS),
                    ("nomfma_nodeq", NO_MFMA + NO_DEQ)):
    s = src
    for old, new in edits:
        assert s.count(old) == 1, (name, old[:40])
        s = s.replace(old, new)
    open(f"/tmp/abl/gemm_{name}.hip", "w").write(s)
    print("wrote", f"/tmp/abl/gemm_{name}.hip")
