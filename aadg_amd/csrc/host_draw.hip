// Host-side planner, part 2 (no GPU work): every draw one training batch makes from PYTHON's `random` generator, in the reference
// pipeline's order -- per (item, domain): per policy the CutMix-queue draw and the sub-policy draw (reference data/policy.py:17-23), then
// DGRandomScaleCrop for the original and each augmented image (data/transform.py:104-131, RandomCrop.draw :38-53), then ToTensor's soft domain
// code (:260-274).  aadg_amd/data/transform.py: _draw_python_stream is the Python statement of the same loop (tests/test_host_cpu.py
// compares the two draw for draw, generator end state included); this one runs it on a COPY of the interpreter's Mersenne-Twister state
// (random.getstate() / setstate()) in ~10 us instead of ~230 us of interpreter time per 24-sample batch.
//
// CPython's generator, restated (Modules/_randommodule.c, Lib/random.py):
//   genrand_uint32()            MT19937 (Matsumoto & Nishimura 2002 reference code)
//   random()                    a = genrand_uint32() >> 5, b = genrand_uint32() >> 6;  (a * 67108864.0 + b) * (1.0 / 9007199254740992.0)
//   getrandbits(k), k <= 32     genrand_uint32() >> (32 - k)
//   _randbelow(n)               k = n.bit_length(); r = getrandbits(k); while r >= n: r = getrandbits(k)
//   randint(a, b) = a + _randbelow(b - a + 1);  uniform(a, b) = a + (b - a) * random()
#include "common.h"

namespace {

struct PyMT {
    uint32_t* mt;      // [624] state words, [624] = position
    uint32_t next() {
        uint32_t& mti = mt[624];
        if (mti >= 624) {
            static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
            int kk;
            uint32_t y;
            for (kk = 0; kk < 624 - 397; kk++) {
                y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1u];
            }
            for (; kk < 623; kk++) {
                y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u];
            }
            y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 1u];
            mti = 0;
        }
        uint32_t y = mt[mti++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double random() {
        const uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
    }
    static int bit_length(uint32_t n) { int k = 0; while (n) { ++k; n >>= 1; } return k; }
    uint32_t randbelow(uint32_t n) {          // n >= 1
        const int k = bit_length(n);
        uint32_t r = next() >> (32 - k);
        while (r >= n) r = next() >> (32 - k);
        return r;
    }
};

}  // namespace

/* mt_state [625] uint32 (random.getstate()[1]) is advanced in place; queue_lens [M] (the policies' CutMix queue lengths) likewise.
 * Outputs: sub [S * M] int64 = the sub-policy drawn per (sample, policy); geo [(S + S * M) * 5] int32 = (scaled_w, scaled_h, pad, crop_x,
 * crop_y) per row, rows [0, S) the un-augmented images, row S + s * M + j the augmented ones; codes [S * n_code] float64 = the soft domain
 * codes.  S = n_items * D.  AADG_E_BADARG: a bad size, or an empty crop range (where python's randint raises ValueError). */
extern "C" int aadg_draw_python_stream(uint32_t* mt_state, int n_items, int D, int M, const int32_t* nsub, int32_t* queue_lens,
                                       double scale_lo, double scale_hi, int crop_h, int crop_w, int crop_pad, int n_code, int W0, int H0,
                                       int64_t* sub, int32_t* geo, double* codes) {
    if (mt_state == nullptr || nsub == nullptr || queue_lens == nullptr || sub == nullptr || geo == nullptr || codes == nullptr)
        return AADG_E_BADARG;
    if (n_items <= 0 || D <= 0 || M <= 0 || n_code < D || W0 <= 0 || H0 <= 0 || crop_h <= 0 || crop_w <= 0 || mt_state[624] > 624)
        return AADG_E_BADARG;
    for (int j = 0; j < M; ++j)
        if (nsub[j] <= 0 || queue_lens[j] < 0) return AADG_E_BADARG;
    PyMT g{mt_state};
    const double ds = scale_hi - scale_lo;
    const int S = n_items * D;
    int s = 0;
    for (int it = 0; it < n_items; ++it)
        for (int d = 0; d < D; ++d, ++s) {
            const int base = S + s * M;
            for (int j = 0; j < M; ++j) {
                int nq = queue_lens[j] + 1;                       // q.append(sample)
                if (nq > 10) nq = 10;                             // q.pop(0): no draw
                else (void)g.randbelow((uint32_t)nq);             // the CutMix-queue draw (its value is unused)
                queue_lens[j] = nq;
                sub[(size_t)s * M + j] = (int64_t)g.randbelow((uint32_t)nsub[j]);
            }
            int row = s;
            for (int jj = 0; jj < M + 1; ++jj) {
                int w = W0, h = H0;
                if (g.random() > 0.2) {
                    w = (int)((scale_lo + ds * g.random()) * W0);
                    h = (int)((scale_lo + ds * g.random()) * H0);
                }
                int pad = 0, pw = w, ph = h;
                if (crop_pad > 0 || w < crop_h || h < crop_w) {
                    // int(max(crop_pad, (crop_h - w) // 2 + 5, (crop_w - h) // 2 + 5)): python's floor division
                    auto fdiv2 = [](int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); };
                    int p = crop_pad;
                    const int a = fdiv2(crop_h - w) + 5, b = fdiv2(crop_w - h) + 5;
                    if (a > p) p = a;
                    if (b > p) p = b;
                    pad = p;
                    pw = w + 2 * pad; ph = h + 2 * pad;
                }
                int x1 = 0, y1 = 0;
                if (!(pw == crop_w && ph == crop_h)) {
                    const int m = pw - crop_w + 1, m2 = ph - crop_h + 1;
                    if (m <= 0 || m2 <= 0) return AADG_E_BADARG;
                    x1 = (int)g.randbelow((uint32_t)m);
                    y1 = (int)g.randbelow((uint32_t)m2);
                }
                int32_t* q = geo + 5 * (size_t)row;
                q[0] = w; q[1] = h; q[2] = pad; q[3] = x1; q[4] = y1;
                row = base + jj;
            }
            double* code = codes + (size_t)s * n_code;
            for (int i = 0; i < n_code; ++i) code[i] = 0.0;
            double used = 0.8 + g.random() * 0.2;
            code[d] = used;
            for (int i = 0; i < n_code; ++i)
                if (i != d) {
                    if (i == n_code - 1) code[i] = 1 - used;
                    else {
                        const double t = g.random() * (1 - used);
                        code[i] = t;
                        used += t;
                    }
                }
        }
    return 0;
}
