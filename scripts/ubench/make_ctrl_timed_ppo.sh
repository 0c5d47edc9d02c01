# Builds scripts/ubench/libaadg_timed.so: libaadg_hip.so with wall_clock64 stamps of workgroup 0 / thread 0 inside k_ctrl_ppo and its
# gradient phase (after every barrier, every counter wait and every `// STAMP` comment), kept in LDS behind both layouts and copied out at
# the end of the launch; extra export aadg_debug_ctrl_times.  Used by scripts/ubench/ctrl_phase_times.py (AADG_LIB_PATH=...).
set -e
cd "$(dirname "$0")/../.."
python - <<'PY'
s = open('aadg_amd/csrc/controller.hip').read()
a = s.index('// The gradient + Adam phase of k_ctrl_ppo as a function of its own')
b = s.index('inline bool ctrl_ok')
head, body, tail = s[:a], s[a:b], s[b:]
out, n, names = [], 0, []
for l in body.split('\n'):
    out.append(l)
    t = l.strip()
    if t == '__syncthreads();' or t.startswith('lds_barrier();') or t.startswith('ppo_arrive_wait(') or t.startswith('if (it + 1 < n_updates) ppo_arrive_wait(') \
       or t.startswith('// STAMP') or t.startswith('ppo_gradients_adam<'):
        n += 1
        names.append((n, t[:70]))
        out.append('    CTT(%d);' % n)
body = '\n'.join(out)
body = body.replace('    for (int it = 0; it < n_updates; ++it) {\n        // the thread index is made opaque per epoch',
                    '    if (threadIdx.x == 0 && blockIdx.x == 0) CTT_BUF[0] = 0;\n    CTT(0);\n    for (int it = 0; it < n_updates; ++it) {\n        // the thread index is made opaque per epoch', 1)
body = body.replace('    // the last workgroup to leave zeroes the counters for the next launch',
                    '    if (threadIdx.x == 0 && blockIdx.x == 0) { const int c = (int)CTT_BUF[0]; for (int i = 0; i < 2 * c; ++i) g_ctt[i] = CTT_BUF[1 + i]; g_ctn = c; }\n'
                    '    // the last workgroup to leave zeroes the counters for the next launch', 1)
head = head.replace('namespace {\n', 'namespace {\n__device__ unsigned long long g_ctt[1024];\n__device__ int g_ctn;\n'
                    '#define CTT_BUF (reinterpret_cast<unsigned long long*>(L + ppo_ga_lds(d).PT + 64))\n'
                    '#define CTT(id) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const int c_ = (int)CTT_BUF[0]; if (c_ < 500) { '
                    'CTT_BUF[1 + 2 * c_] = wall_clock64(); CTT_BUF[2 + 2 * c_] = id; CTT_BUF[0] = c_ + 1; } } } while (0)\n', 1)
head = head.replace('    l.PT = o; o += 2 * 28;', '    l.PT = o; o += 2 * 28 + 8 + 2 * 1024;')
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
open('scripts/ubench/ctrl_stamp_names.txt', 'w').write('\n'.join('%3d  %s' % x for x in names) + '\n')
PY
mkdir -p /tmp/tl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iaadg_amd/csrc -c /tmp/controller_timed.hip -o /tmp/tl/controller.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/ubench/libaadg_timed.so /tmp/tl/controller.o $(ls aadg_amd/lib/obj/*.o | grep -v controller.o)
