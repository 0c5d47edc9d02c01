# Builds scripts/ubench/libaadg_timed.so: libaadg_hip.so with a wall_clock64 stamp after every barrier of k_ctrl_rollout
# (workgroup 0) and an extra export aadg_debug_ctrl_times; used by scripts/ubench/ctrl_phase_times.py.
set -e
cd "$(dirname "$0")/../.."
python - <<'PY'
s = open('aadg_amd/csrc/controller.hip').read()
a = s.index('__global__ __launch_bounds__(CT_THREADS) void k_ctrl_rollout')
b = s.index('// sum_r a[r * sa] * b[r * sb]')
head, body, tail = s[:a], s[a:b], s[b:]
out, n = [], 0
for l in body.split('\n'):
    out.append(l)
    if l.strip() == '__syncthreads();':
        n += 1
        out.append('    CTT(%d);' % n)
body = '\n'.join(out)
body = body.replace('    extern __shared__ __attribute__((aligned(16))) float L[];\n',
                    '    extern __shared__ __attribute__((aligned(16))) float L[];\n    int ctn = 0;\n    CTT(0);\n', 1)
head = head.replace('namespace {\n', 'namespace {\n__device__ unsigned long long g_ctt[1024];\n__device__ int g_ctn;\n'
                    '#define CTT(id) do { if (threadIdx.x == 0 && blockIdx.x == 0 && ctn < 500) { g_ctt[2 * ctn] = wall_clock64(); '
                    'g_ctt[2 * ctn + 1] = id; ++ctn; g_ctn = ctn; } } while (0)\n', 1)
tail += '''
extern "C" int aadg_debug_ctrl_times(unsigned long long* out, int cap) {
    int n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_ctn), sizeof(int)) != hipSuccess) return -1;
    if (n > cap) n = cap;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ctt), sizeof(unsigned long long) * 2 * n) != hipSuccess) return -1;
    return n;
}
'''
open('/tmp/controller_timed.hip', 'w').write(head + body + tail)
PY
mkdir -p /tmp/tl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Iaadg_amd/csrc -c /tmp/controller_timed.hip -o /tmp/tl/controller.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/ubench/libaadg_timed.so /tmp/tl/controller.o $(ls aadg_amd/lib/obj/*.o | grep -v controller.o)
