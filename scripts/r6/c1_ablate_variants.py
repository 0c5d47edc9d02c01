"""Timing-only variants of the f32x3 1x1 kernel (results WRONG): which phase of its K loop costs what.  Builds exp_libs/{noglobal,nomfma,nosplit,onemfma}.so
from csrc/conv1x1_fwd.hip with one phase removed in the EXACT float32 instantiation; run them with scripts/r6/c1_ablate.sh.
    python scripts/r6/c1_ablate_variants.py && gpurun -- 'bash scripts/r6/c1_ablate.sh noglobal nomfma nosplit onemfma'
Round 6 (144 images, ms): shape            tree   noglobal nomfma nosplit onemfma
                          512->2048 @32^2  1.28-1.31  0.97  0.83   1.24    0.89      (MFMA floor 0.37)
                          2048->512 @32^2  1.12       0.76  0.73   1.08    0.75
                          256->1024 @32^2  0.31       0.22  0.24   0.30    0.26
                          2048->256 @32^2  0.60       0.39  0.42   0.56    0.44
                          64->256 @128^2   0.55       0.48  0.50   0.53    0.51      (HBM-bound: floor 0.48)
=> the data pipeline WITHOUT any MFMA takes 65 % of the kernel, the global loads a quarter, the split VALU 3-5 %: the three phases of a K-step
(global -> registers, split + LDS write + barrier, fragment reads + MFMA) run one after the other instead of beside each other."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aadg_amd import build as b
src = open(os.path.join(b.CSRC, "conv1x1_fwd.hip")).read()
MF = """                    if (X3) {                              // the small cross terms first, the hi * hi product last
                        d[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL - 1][mi], b[0][ni], d[mi][ni], 0, 0, 0);
                        d[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[PL - 1][ni], d[mi][ni], 0, 0, 0);
                    }
                    d[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[0][ni], d[mi][ni], 0, 0, 0);"""
V = {
    "noglobal": [("        if (k0 + BK < K) fetch(k0 + BK);                  // in flight during the MFMAs below",
                  "        if (X3 && EXACT) { asm volatile(\"\" ::: \"memory\"); } else if (k0 + BK < K) fetch(k0 + BK);")],
    "nomfma": [(MF, """                    if (X3 && EXACT) {
                        const uint4 ua = __builtin_bit_cast(uint4, a[0][mi]), ub = __builtin_bit_cast(uint4, b[0][ni]), uc = __builtin_bit_cast(uint4, a[PL - 1][mi]), ud = __builtin_bit_cast(uint4, b[PL - 1][ni]);
                        d[mi][ni][0] += __uint_as_float(ua.x ^ ub.y ^ uc.z ^ ud.w ^ ua.w ^ ub.x ^ uc.y ^ ud.z ^ ua.y ^ ua.z ^ ub.z ^ ub.w ^ uc.x ^ uc.w ^ ud.x ^ ud.y);
                    } else {
""" + MF + "\n                    }")],
    "nosplit": [("                aadg_split4(f, hi, lo);",
                 "                if (EXACT) { hi = make_uint2(rb[i].x, rb[i].y); lo = make_uint2(rb[i].z, rb[i].w); } else aadg_split4(f, hi, lo);")],
    "onemfma": [("                    if (X3) {                              // the small cross terms first, the hi * hi product last",
                 "                    if (X3 && !EXACT) {")],
}
b.build_hip()
os.makedirs(os.path.join(ROOT, "exp_libs"), exist_ok=True)
objdir = os.path.join(b.LIB_DIR, "obj")
for tag, edits in V.items():
    s = src
    for old, new in edits:
        assert old in s, tag
        s = s.replace(old, new)
    path = "/tmp/c1_%s.hip" % tag
    open(path, "w").write(s)
    obj = "/tmp/_variant_%s.o" % tag
    subprocess.check_call([b._hipcc()] + b.HIPCC_FLAGS + ["-c", path, "-o", obj])
    objs = [obj if o == "conv1x1_fwd.o" else os.path.join(objdir, o) for o in sorted(os.listdir(objdir)) if o.endswith(".o")]
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "exp_libs", tag + ".so")] + objs)
    print(tag)
