"""Timing-only variants of k_big_cost_x3 (results WRONG): what the large-cloud cost build spends its 0.41 ms on.
    python scripts/r6/skbig_ablate.py && gpurun -- 'for t in tree nostore notrans nomfma noload rowstore; do echo -n "$t: "; AADG_LIB_PATH=exp_libs/skb_$t.so PYTHONPATH=scripts/ab/hook python scripts/r6/skbig_phases.py 2>/dev/null | tail -1; done'
(`tree` = the library as it is.)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aadg_amd import build as b
src = open(os.path.join(b.CSRC, "sinkhorn_big.hip")).read()
a0 = src.index("void k_big_cost_x3(")
a1 = src.index("// ---- one sweep:")
body = src[a0:a1]
STORE = "if (row < rows && col < cols) C[(size_t)row * nmax + col] = 1.0f - acc[a][b][r];"
TRANS = "if (row < rows && col < cols) Ct[(size_t)col * nmax + row] = T[rr * 33 + c];"
MFMA = "acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16("
LOAD = "if (gr < (isb ? cols : rows) && k < E) v = *reinterpret_cast<const float4*>((isb ? Bm : Am) + (size_t)gr * E + k);"

NEW_EPILOGUE = """    // the whole 128 x 128 tile through LDS (the operand buffers, exactly 128 x 136 floats): a wave instruction then writes two
    // full 512-byte rows of the tile as float4 instead of 2 x 128 bytes as dwords
    constexpr int TP = 136;
    float* T = reinterpret_cast<float*>(lds_x3);
    const bool vec = (nmax & 3) == 0;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                T[(wi + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * g) * TP + wj + 32 * b + (lane & 31)] = 1.0f - acc[a][b][r];
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int r = 8 * it + (tid >> 5), c = 4 * (tid & 31), row = i0 + r, col = j0 + c;
        if (row >= rows || col >= cols) continue;
        const float4 v = *reinterpret_cast<const float4*>(T + r * TP + c);
        float* o = C + (size_t)row * nmax + col;
        if (vec && col + 3 < cols) *reinterpret_cast<float4*>(o) = v;
        else { o[0] = v.x; if (col + 1 < cols) o[1] = v.y; if (col + 2 < cols) o[2] = v.z; if (col + 3 < cols) o[3] = v.w; }
    }
    if (which == 2) {
        // C_yx = C_xy^T: the tile transposed on its way into LDS, then the same row stores
        float* Ct = base + L.cyx;
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    T[(wj + 32 * b + (lane & 31)) * TP + wi + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * g] = 1.0f - acc[a][b][r];
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int r = 8 * it + (tid >> 5), c = 4 * (tid & 31), row = j0 + r, col = i0 + c;       // row of C_yx = column of C_xy
            if (row >= cols || col >= rows) continue;
            const float4 v = *reinterpret_cast<const float4*>(T + r * TP + c);
            float* o = Ct + (size_t)row * nmax + col;
            if (vec && col + 3 < rows) *reinterpret_cast<float4*>(o) = v;
            else { o[0] = v.x; if (col + 1 < rows) o[1] = v.y; if (col + 2 < rows) o[2] = v.z; if (col + 3 < rows) o[3] = v.w; }
        }
    }
}

"""
e0 = body.index("    // C/D layout: col = lane & 31")
V = {
    "rowstore": [(body[e0:], NEW_EPILOGUE)],
    "tree": [],
    "nostore": [(STORE, "if (row < rows && col < cols && acc[a][b][r] == 12345.678f) C[(size_t)row * nmax + col] = 1.0f - acc[a][b][r];"),
                (TRANS, "if (row < rows && col < cols && T[rr * 33 + c] == 12345.678f) Ct[(size_t)col * nmax + row] = T[rr * 33 + c];")],
    "notrans": [("    if (which == 2) {\n        // C_yx = C_xy^T, transposed through LDS (the operand buffers are free)", "    if (which == 7) {\n        // C_yx = C_xy^T, transposed through LDS (the operand buffers are free)")],
    "nomfma": [(MFMA, "if (k0 > E) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(")],
    "noload": [(LOAD, "if (gr < (isb ? cols : rows) && k < E && E == 7) v = *reinterpret_cast<const float4*>((isb ? Bm : Am) + (size_t)gr * E + k);")],
}
b.build_hip()
os.makedirs(os.path.join(ROOT, "exp_libs"), exist_ok=True)
objdir = os.path.join(b.LIB_DIR, "obj")
for tag, edits in V.items():
    s = body
    for old, new in edits:
        assert old in s, (tag, old)
        s = s.replace(old, new)
    path = "/tmp/skb_%s.hip" % tag
    open(path, "w").write(src[:a0] + s + src[a1:])
    obj = "/tmp/_skb_%s.o" % tag
    subprocess.check_call([b._hipcc()] + b.HIPCC_FLAGS + ["-I", b.CSRC, "-c", path, "-o", obj])
    objs = [obj if o == "sinkhorn_big.o" else os.path.join(objdir, o) for o in sorted(os.listdir(objdir)) if o.endswith(".o")]
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "exp_libs", "skb_%s.so" % tag)] + objs)
    print(tag)
