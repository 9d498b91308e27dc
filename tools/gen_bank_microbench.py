#!/usr/bin/env python3
"""Writes a HIP microbenchmark (stdout) that measures how the VGPR bank (register number mod 4) of every
operand of v_mul_f32_e32 / v_fmac_f32_e32 / v_fma_f32 changes the issue rate on gfx950.  Each kernel is 32
independent instructions in inline assembly with explicit registers: four destinations in one bank in
rotation, sources fixed.  Usage: gen_bank_microbench.py > x.hip && hipcc --offload-arch=gfx950 -O3 x.hip"""

def reg(bank, k):  # k-th register of a bank, from v16 up
    return 16 + 4 * k + bank

cases = []
for op in ("mul", "fmac", "fma"):
    nsrc = 3 if op == "fma" else 2
    combos = set()
    for d in range(2):
        for a in range(3):
            for b in range(3):
                if nsrc == 2:
                    combos.add((d, a, b))
                else:
                    for c in range(3):
                        combos.add((d, a, b, c))
    # canonical subset: keep the patterns up to renaming of banks
    seen = set()
    for combo in sorted(combos):
        ren, key = {}, []
        for x in combo:
            ren.setdefault(x, len(ren))
            key.append(ren[x])
        key = tuple(key)
        if key in seen:
            continue
        seen.add(key)
        cases.append((op, key))

print("#include <hip/hip_runtime.h>\n#include <cstdio>")
print("#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf(\"HIP error %s line %d\\n\", hipGetErrorString(e_), __LINE__); return 1; } } while (0)")
clob = ",".join(f'"v{r}"' for r in range(16, 80))
names = []
for op, key in cases:
    d = key[0]
    srcs = key[1:]
    # distinct registers: sources take slots 8.. of their bank, one slot each
    used, sreg = {}, []
    for b in srcs:
        used[b] = used.get(b, 0) + 1
        sreg.append(reg(b, 7 + used[b]))
    body = ""
    for i in range(32):
        dr = reg(d, i % 4)
        if op == "mul":
            body += f"v_mul_f32_e32 v{dr}, v{sreg[0]}, v{sreg[1]}\\n"
        elif op == "fmac":
            body += f"v_fmac_f32_e32 v{dr}, v{sreg[0]}, v{sreg[1]}\\n"
        else:
            body += f"v_fma_f32 v{dr}, v{sreg[0]}, v{sreg[1]}, v{sreg[2]}\\n"
    name = f"k_{op}_d{d}_" + "".join(str(b) for b in srcs)
    names.append((name, f"{op} dst bank {d} src banks {','.join(map(str, srcs))}"))
    print(f"__global__ void __launch_bounds__(256) {name}(long long* ticks, int iters) {{")
    print("    const long long c0 = clock64(), w0 = wall_clock64();")
    print(f"    for (int i = 0; i < iters; i++) asm volatile(\"{body}\" ::: {clob});")
    print("    const long long c1 = clock64(), w1 = wall_clock64();")
    print("    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = c1 - c0; ticks[1] = w1 - w0; }\n}")
# explicit patterns: (label, op, destination registers in rotation, source registers)
explicit = [
    ("fmac dst v16..v23 src v32,v36", "fmac", list(range(16, 24)), [32, 36]),
    ("fmac dst banks 1,2,3 only src v32,v36", "fmac", [17, 18, 19, 21, 22, 23], [32, 36]),
    ("fmac dst v16..v23 src v32,v37", "fmac", list(range(16, 24)), [32, 37]),
    ("fmac dst v16,v20,v24,v28 src v33,v37", "fmac", [16, 20, 24, 28], [33, 37]),
    ("fmac dst v17,v21,v25,v29 src v32,v36", "fmac", [17, 21, 25, 29], [32, 36]),
    ("fma dst v16..v23 src v32,v36,v33", "fma", list(range(16, 24)), [32, 36, 33]),
    ("fma dst v16..v23 src v32,v36,v40", "fma", list(range(16, 24)), [32, 36, 40]),
    ("mul dst v16..v23 src v32,v36", "mul", list(range(16, 24)), [32, 36]),
    ("fmac dst v16..v23 src v32,v36 then mul v33,v34 alternating", "fmac+mul", list(range(16, 24)), [32, 36]),
]
for idx, (label, op, dsts, srcs) in enumerate(explicit):
    body = ""
    for i in range(32):
        dr = dsts[i % len(dsts)]
        if op == "mul" or (op == "fmac+mul" and i % 2 == 1):
            body += f"v_mul_f32_e32 v{dr}, v{33 if op != 'mul' else srcs[0]}, v{34 if op != 'mul' else srcs[1]}\\n"
        elif op.startswith("fmac"):
            body += f"v_fmac_f32_e32 v{dr}, v{srcs[0]}, v{srcs[1]}\\n"
        else:
            body += f"v_fma_f32 v{dr}, v{srcs[0]}, v{srcs[1]}, v{srcs[2]}\\n"
    name = f"k_explicit_{idx}"
    names.append((name, label))
    print(f"__global__ void __launch_bounds__(256) {name}(long long* ticks, int iters) {{")
    print("    const long long c0 = clock64(), w0 = wall_clock64();")
    print(f"    for (int i = 0; i < iters; i++) asm volatile(\"{body}\" ::: {clob});")
    print("    const long long c1 = clock64(), w1 = wall_clock64();")
    print("    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = c1 - c0; ticks[1] = w1 - w0; }\n}")
print("""
template <typename K>
static int run(const char* name, K kernel, long long* d_ticks) {
    for (int waves_per_simd : {2, 4}) {
        const int blocks = 256 * waves_per_simd, iters = 20000;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_ticks, 200);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_ticks, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        long long t[2];
        CK(hipMemcpy(t, d_ticks, sizeof t, hipMemcpyDeviceToHost));
        const double mhz = 100.0 * t[0] / t[1];
        const double ns = ms * 1e6 / ((double)waves_per_simd * iters * 32);
        printf("%-60s waves/SIMD=%d  %5.0f MHz  %5.2f cycles/instr\\n", name, waves_per_simd, mhz, ns * mhz * 1e-3);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    return 0;
}
int main() {
    long long* d_ticks;
    CK(hipMalloc(&d_ticks, 64));""")
for name, label in names:
    print(f'    if (run("{label}", {name}, d_ticks)) return 1;')
print('    printf("done\\n");\n    return 0;\n}')
