# generates tools/ubench/valu_issue.hip
ops = []  # (name, kind, mnemonic)
def add(kind, *mn):
    for m in mn: ops.append((m, kind, m))
add("F3", "v_fma_f32", "v_max3_f32", "v_min3_f32", "v_med3_f32", "v_mad_u32_u24", "v_lshl_add_u32", "v_add3_u32", "v_and_or_b32",
    "v_bfe_u32", "v_perm_b32", "v_fma_f64x")
add("F2", "v_min_f32", "v_max_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32", "v_xor_b32",
    "v_mul_lo_u32", "v_mul_u32_u24", "v_min_u32", "v_max_u32", "v_mul_hi_u32")
add("F2R", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32")
add("FMAC", "v_fmac_f32")
add("F1", "v_mov_b32", "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte1", "v_cvt_f32_u32", "v_cvt_u32_f32", "v_cvt_f32_i32", "v_rcp_f32", "v_sqrt_f32",
    "v_rsq_f32", "v_not_b32", "v_bfrev_b32")
add("CMPVCC", "v_cmp_le_f32", "v_cmp_eq_u32", "v_cmp_lt_f32")
add("CMPS", "v_cmp_le_f32_e64")
add("CNDS", "v_cndmask_b32_e64")
add("CNDVCC", "v_cndmask_b32")
add("PK3", "v_pk_fma_f32")
add("PK2", "v_pk_mul_f32", "v_pk_add_f32")
add("DEP", "v_fma_f32 (one dependent chain)")
add("MIX", "node-step mix: 2 cvt_ubyte, pk_fma, max3, 2 min, cmp, cndmask")
add("MAD64", "v_mad_u64_u32")
add("READLANE", "v_readfirstlane_b32")
add("MIXF", "v_fma_mix_f32 (f32 operands)", "v_fma_mix_f32 (src0 = low f16)", "v_fma_mix_f32 (src0 = high f16)")
add("F1", "v_cvt_f32_f16")
add("F3", "v_pk_fma_f16")
add("F2", "v_pk_min_f16", "v_pk_max_f16", "v_pk_mul_f16")
add("SDWA", "v_mul_f32_sdwa (src0 = byte 1)")
# gfx950's narrow-float converts (r03: could the node planes be decoded two per instruction?)
add("CVTPK", "v_cvt_pk_f32_fp8", "v_cvt_pk_f32_bf8", "v_cvt_pk_f32_fp8_sdwa (src0 = word 1)", "v_cvt_scalef32_pk_f32_fp8", "v_cvt_scalef32_pk_f32_fp8 (op_sel word 1)",
    "v_cvt_scalef32_pk_f32_fp4")
add("F1", "v_cvt_f32_fp8")
add("CVT32", "v_cvt_scalef32_pk32_f32_fp6")
add("DSW", "ds_write_b32", )
add("DSR", "ds_read_b32", )

def body(kind, mn):
    regs8 = ', '.join(f'"+v"(a[{i}])' for i in range(8))
    def rep(fmt, n=8):
        return '"' + '\\n "\n        "'.join(fmt.replace('K', str(k)) for k in range(n)) + '"'
    if kind == "F3":
        if mn == "v_fma_f64x":
            return ('    double d0 = a[0], d1 = a[1], d2 = a[2], d3 = a[3], db = b, dc = c;\n'
                    '    REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\\n v_fma_f64 %1, %1, %4, %5\\n v_fma_f64 %2, %2, %4, %5\\n v_fma_f64 %3, %3, %4, %5\\n"\n'
                    '        "v_fma_f64 %0, %0, %4, %5\\n v_fma_f64 %1, %1, %4, %5\\n v_fma_f64 %2, %2, %4, %5\\n v_fma_f64 %3, %3, %4, %5"\n'
                    '        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));)\n'
                    '    a[0] = (float)d0; a[1] = (float)d1; a[2] = (float)d2; a[3] = (float)d3;\n')
        return f'    REP8(asm volatile({rep(mn + " %K, %K, %8, %9")} : {regs8} : "v"(b), "v"(c));)\n'
    if kind == "MIXF":
        sel = " op_sel_hi:[1,0,0]" if "low f16" in mn else (" op_sel:[1,0,0] op_sel_hi:[1,0,0]" if "high f16" in mn else "")
        return f'    REP8(asm volatile({rep("v_fma_mix_f32 %K, %8, %9, %K" + sel)} : {regs8} : "v"(u), "v"(c));)\n'
    if kind == "SDWA":
        return f'    REP8(asm volatile({rep("v_mul_f32_sdwa %K, %8, %K dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")} : {regs8} : "v"(u));)\n'
    if kind == "F2":
        return f'    REP8(asm volatile({rep(mn + " %K, %K, %8")} : {regs8} : "v"(b));)\n'
    if kind == "F2R":
        return f'    REP8(asm volatile({rep(mn + " %K, 1, %K")} : {regs8});)\n'
    if kind == "FMAC":
        return f'    REP8(asm volatile({rep(mn + " %K, %8, %9")} : {regs8} : "v"(b), "v"(c));)\n'
    if kind == "F1":
        return f'    REP8(asm volatile({rep(mn + " %K, %8")} : {regs8} : "v"(u));)\n'
    if kind == "CMPVCC":
        return f'    REP8(asm volatile({rep(mn + " vcc, %K, %8")} : {regs8} : "v"(b) : "vcc");)\n'
    if kind == "CMPS":
        return ('    unsigned long long m = 0;\n'
                f'    REP8(asm volatile({rep("v_cmp_le_f32_e64 %8, %K, %9")} : {regs8}, "+s"(m) : "v"(b));)\n'
                '    a[0] += (float)(uint32_t)m;\n')
    if kind == "CNDS":
        return ('    const unsigned long long m = 0x5555aaaa3333ccccull ^ u;\n'
                f'    REP8(asm volatile({rep("v_cndmask_b32_e64 %K, %K, %8, %9")} : {regs8} : "v"(b), "s"(m));)\n')
    if kind == "CNDVCC":
        return ('    asm volatile("v_cmp_le_f32 vcc, %0, %1" :: "v"(b), "v"(c) : "vcc");\n'
                f'    REP8(asm volatile({rep("v_cndmask_b32 %K, %K, %8, vcc")} : {regs8} : "v"(b) : "vcc");)\n')
    if kind in ("PK3", "PK2"):
        tail = ", %4, %5" if kind == "PK3" else ", %4"
        ins = ' "\n        "'.join(f'{mn} %{k % 4}, %{k % 4}{tail}\\n' for k in range(8))
        return ('    f2 p0 = {a[0], a[1]}, p1 = {a[2], a[3]}, p2 = {a[4], a[5]}, p3 = {a[6], a[7]}, bb = {b, b}, cc = {c, c};\n'
                f'    REP8(asm volatile("{ins}" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb), "v"(cc));)\n'
                '    a[0] = p0.x; a[1] = p0.y; a[2] = p1.x; a[3] = p1.y; a[4] = p2.x; a[5] = p2.y; a[6] = p3.x; a[7] = p3.y;\n')
    if kind == "CVTPK":
        base = mn.split(" ")[0]
        if "sdwa" in mn:
            fmt = base + " %{r}, %4 src0_sel:WORD_1"
        elif "scalef32" in mn:
            fmt = base + " %{r}, %4, %5" + (" op_sel:[1,0,0]" if "op_sel" in mn else "")
        else:
            fmt = base + " %{r}, %4"
        ins = ' "\n        "'.join(fmt.replace("{r}", str(k % 4)) + '\\n' for k in range(8))
        return ('    f2 p0 = {a[0], a[1]}, p1 = {a[2], a[3]}, p2 = {a[4], a[5]}, p3 = {a[6], a[7]};\n'
                f'    REP8(asm volatile("{ins}" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(u), "v"(b));)\n'
                '    a[0] = p0.x; a[1] = p0.y; a[2] = p1.x; a[3] = p1.y; a[4] = p2.x; a[5] = p2.y; a[6] = p3.x; a[7] = p3.y;\n')
    if kind == "CVT32":  # 32 results per instruction: 8 instructions write the same 32 registers from 6 source registers
        return ('    typedef float f32v __attribute__((ext_vector_type(32)));\n    typedef uint32_t u6v __attribute__((ext_vector_type(6)));\n'
                '    f32v big; for (int i = 0; i < 32; i++) big[i] = a[i & 7];\n    u6v src = {u, u + 1, u + 2, u + 3, u + 4, u + 5};\n'
                '    REP8(asm volatile("v_cvt_scalef32_pk32_f32_fp6 %0, %1, %2\\n v_cvt_scalef32_pk32_f32_fp6 %0, %1, %2\\n v_cvt_scalef32_pk32_f32_fp6 %0, %1, %2\\n v_cvt_scalef32_pk32_f32_fp6 %0, %1, %2\\n"\n'
                '        "v_cvt_scalef32_pk32_f32_fp6 %0, %1, %2\\n v_cvt_scalef32_pk32_f32_fp6 %0, %1, %2\\n v_cvt_scalef32_pk32_f32_fp6 %0, %1, %2\\n v_cvt_scalef32_pk32_f32_fp6 %0, %1, %2" : "+v"(big) : "v"(src), "v"(b));)\n'
                '    for (int i = 0; i < 32; i++) a[i & 7] += big[i];\n')
    if kind == "DEP":
        return f'    REP8(asm volatile({rep("v_fma_f32 %0, %0, %8, %9")} : {regs8} : "v"(b), "v"(c));)\n'
    if kind == "MIX":
        return ('    f2 p0 = {a[0], a[1]}, bb = {b, b}, cc = {c, c};\n'
                '    REP8(asm volatile("v_cvt_f32_ubyte0 %1, %6\\n v_cvt_f32_ubyte1 %2, %6\\n v_pk_fma_f32 %0, %0, %7, %8\\n v_max3_f32 %3, %1, %2, %3\\n"\n'
                '        "v_min_f32 %4, %4, %2\\n v_min_f32 %5, %5, %1\\n v_cmp_le_f32 vcc, %3, %4\\n v_cndmask_b32 %5, %5, %3, vcc"\n'
                '        : "+v"(p0), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]) : "v"(u), "v"(bb), "v"(cc) : "vcc");)\n'
                '    a[0] = p0.x; a[1] = p0.y;\n')
    if kind == "MAD64":
        ins = ' "\n        "'.join(f'v_mad_u64_u32 %{k % 4}, vcc, %4, %5, %{k % 4}\\n' for k in range(8))
        return ('    unsigned long long q0 = u, q1 = u + 1, q2 = u + 2, q3 = u + 3;\n'
                f'    REP8(asm volatile("{ins}" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(u), "v"(u) : "vcc");)\n'
                '    a[0] += (float)(q0 + q1 + q2 + q3);\n')
    if kind == "READLANE":
        return ('    uint32_t s0 = 0;\n'
                f'    REP8(asm volatile({rep("v_readfirstlane_b32 %8, %K")} : {regs8}, "+s"(s0));)\n'
                '    a[0] += (float)s0;\n')
    if kind == "DSW":
        return ('    const uint32_t addr = (threadIdx.x & 63u) * 4u + (threadIdx.x >> 6) * 256u;\n'
                f'    REP8(asm volatile({rep("ds_write_b32 %8, %K")} "\\n s_waitcnt lgkmcnt(0)" : {regs8} : "v"(addr) : "memory");)\n')
    if kind == "DSR":
        return ('    const uint32_t addr = (threadIdx.x & 63u) * 4u + (threadIdx.x >> 6) * 256u;\n'
                f'    REP8(asm volatile({rep("ds_read_b32 %K, %8")} "\\n s_waitcnt lgkmcnt(0)" : {regs8} : "v"(addr) : "memory");)\n')
    raise KeyError(kind)

out = []
out.append(r'''// GENERATED by tools/ubench/gen_valu_issue.py -- do not edit by hand.
// Microbenchmark: vector-instruction issue cost on one gfx950 SIMD -- cycles per wave-instruction per SIMD as a function
// of the waves resident on the SIMD, the instruction, and the EXEC mask.  It calibrates the VALU roof that render_kernel
// is priced against (DESIGN.md section 6, bench.py roofline): MI355X_MICROARCH.md's constants table gives v_fma_f32 as
// 2 cycles per wave64 instruction on the SIMD-32 with more than one wave resident and 4 for one wave alone; this table
// adds the other instructions of the traversal step, several of which turn out to issue at HALF that rate.
//
//   hipcc -O3 --offload-arch=gfx950 valu_issue.hip -o valu_issue && ./valu_issue
//
// Method: one launch of n_cu * W workgroups of 4 waves, dynamic LDS sized so that a CU holds exactly W of them (W waves
// per SIMD); every wave runs ITERS x 64 instructions (8 independent accumulator registers; inline asm, nothing folded)
// between two s_memtime stamps and records HW_REG_HW_ID / HW_REG_XCC_ID.  Per SIMD (xcc, se, sh, cu, simd) the host
// takes the waves that ran there, the window in which ALL of them were running (max t0 .. min t1) and the instructions
// issued inside it (each wave issues uniformly over its own span): cycles per wave-instruction per SIMD = window /
// instructions in the window.  SIMDs that did not hold exactly W waves (dispatcher imbalance) are left out.\n// All workgroups cross a chip-wide start line first, so the W waves of a SIMD run side by side for their whole span\n// (`overlap` = common window / union of the spans, reported; 1 = perfect); `clock` = d s_memtime / d s_memrealtime.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define REP8(X) X X X X X X X X
typedef float f2 __attribute__((ext_vector_type(2)));
''')
out.append(f"enum {{ N_OPS = {len(ops)} }};\n")
out.append("static const char *kNames[N_OPS] = {" + ", ".join('"%s"' % o[0] for o in ops) + "};\n\n")
out.append("template <int OP>\n__device__ __forceinline__ void body(float (&a)[8], float b, float c, uint32_t u) {\n")
for i, (name, kind, mn) in enumerate(ops):
    out.append(f"  if (OP == {i}) {{  // {name}\n" + body(kind, mn) + "  }\n")
out.append("}\n")
out.append(r'''
struct WaveRec {
  unsigned long long t0, t1;  // s_memtime (shader cycles)
  unsigned long long r0, r1;  // s_memrealtime (100 MHz)
  uint32_t hw_id, xcc_id;     // HW_REG_HW_ID (wave / simd / cu / sh / se), HW_REG_XCC_ID
};
// lanes: 64 = full EXEC; 32 = lanes 0..31 only; 33 = lanes 0..15 and 32..47; 16 = lanes 0..15; 1 = lane 0
template <int OP>
__global__ void __launch_bounds__(256) issue_kernel(int iters, int lanes, float b, float c, uint32_t u, WaveRec *recs, float *sink, uint32_t *start_line) {
  extern __shared__ char lds_pad[];
  float a[8];
  for (int i = 0; i < 8; i++) a[i] = (float)(threadIdx.x + i) * 1e-3f;
  const uint32_t lane = threadIdx.x & 63u;
  const bool on = lanes == 64 || (lanes == 32 && lane < 32u) || (lanes == 33 && (lane & 16u) == 0u) || (lanes == 16 && lane < 16u) || (lanes == 1 && lane == 0u);
  // chip-wide start line: the grid is exactly what the device holds (W workgroups per CU), so every workgroup is
  // resident; bounded spin in case it is not (the host then sees spans that do not coincide and says so)
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(start_line, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0; spin < (1 << 22); spin++) {
      if (__hip_atomic_load(start_line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) break;
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (on) {
    for (int it = 0; it < iters; it++) body<OP>(a, b, c, u);
  }
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; i++) s += a[i];
  if (s == 1234.5678f) sink[0] = s + (float)lds_pad[threadIdx.x];
  if ((threadIdx.x & 63u) == 0u) {
    WaveRec r;
    r.t0 = t0; r.t1 = t1; r.r0 = r0; r.r1 = r1;
    r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
    r.xcc_id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID[3:0]
    recs[blockIdx.x * 4u + (threadIdx.x >> 6)] = r;
  }
}

static WaveRec *d_recs;
static float *sink;
static uint32_t *d_start;
static int n_cu;
static const char *g_only;  // "only <substring>": the sweep runs just the instructions whose name contains it

template <int OP>
double run(int W, int lanes, bool print = true) {
  const int iters = 3000;
  const size_t lds = (160u * 1024u) / (size_t)W - (W == 1 ? 0 : 64);  // W workgroups per CU by LDS
  const int blocks = n_cu * W;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(issue_kernel<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipMemset(d_start, 0, 8);
  hipLaunchKernelGGL(issue_kernel<OP>, dim3(blocks), dim3(256), lds, 0, 50, lanes, 1.0001f, 1e-7f, 0x01020304u, d_recs, sink, d_start);
  hipLaunchKernelGGL(issue_kernel<OP>, dim3(blocks), dim3(256), lds, 0, iters, lanes, 1.0001f, 1e-7f, 0x01020304u, d_recs, sink, d_start + 1);
  (void)hipDeviceSynchronize();
  std::vector<WaveRec> recs((size_t)blocks * 4);
  (void)hipMemcpy(recs.data(), d_recs, recs.size() * sizeof(WaveRec), hipMemcpyDeviceToHost);
  const double insts = (double)iters * 64.0;  // wave-instructions per wave
  std::map<uint32_t, std::vector<const WaveRec *>> by_simd;
  for (const WaveRec &r : recs) by_simd[(r.xcc_id << 16) | (r.hw_id & 0xff30u)].push_back(&r);  // se | sh | cu | simd
  std::vector<double> cyc, solo, overlap, clk;
  int odd = 0;
  for (auto &kv : by_simd) {
    const auto &v = kv.second;
    if ((int)v.size() != W) { odd++; continue; }
    unsigned long long lo = 0, hi = ~0ull;
    double issued = 0;
    for (const WaveRec *r : v) { lo = std::max(lo, r->t0); hi = std::min(hi, r->t1); }
    if (hi <= lo) { odd++; continue; }
    unsigned long long lo0 = ~0ull, hi1 = 0;
    for (const WaveRec *r : v) { lo0 = std::min(lo0, r->t0); hi1 = std::max(hi1, r->t1); }
    overlap.push_back((double)(hi - lo) / (double)(hi1 - lo0));  // 1 = the W waves ran side by side from start to end
    for (const WaveRec *r : v) clk.push_back((double)(r->t1 - r->t0) / (double)(r->r1 - r->r0) * 0.1);  // GHz
    // all W waves issue `insts` instructions inside [lo0, hi1]
    (void)issued;
    cyc.push_back((double)(hi1 - lo0) / (insts * W));
    for (const WaveRec *r : v) solo.push_back((double)(r->t1 - r->t0) / insts);
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(solo.begin(), solo.end());
  std::sort(overlap.begin(), overlap.end());
  std::sort(clk.begin(), clk.end());
  if (cyc.empty()) {
    if (print) printf("%-62s W %d lanes %2d  NO SIMD held exactly %d waves (%zu SIMDs seen)\n", kNames[OP], W, lanes, W, by_simd.size());
    return 0;
  }
  if (print)
    printf("%-62s W %d lanes %2d  %6.3f cyc/instr/SIMD  %7.3f cyc/instr per wave  overlap %.3f (min %.3f)  clock %.2f GHz  (%zu SIMDs, %d left out)\n",
           kNames[OP], W, lanes, cyc[cyc.size() / 2], solo[solo.size() / 2], overlap[overlap.size() / 2], overlap[0], clk[clk.size() / 2], cyc.size(), odd);
  return cyc[cyc.size() / 2];
}

template <int OP>
struct Sweep {
  static void go(const int *Ws, int nW, double *table) {
    for (int i = 0; i < nW && (!g_only || strstr(kNames[OP], g_only)); i++) table[OP * nW + i] = run<OP>(Ws[i], 64);
    Sweep<OP + 1>::go(Ws, nW, table);
  }
};
template <>
struct Sweep<N_OPS> {
  static void go(const int *, int, double *) {}
};

int main(int argc, char **argv) {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  n_cu = p.multiProcessorCount;
  printf("# %s, %d CUs; cycles are s_memtime ticks (shader clock); W = waves resident per SIMD\n", p.gcnArchName, n_cu);
  (void)hipMalloc(&d_recs, (size_t)n_cu * 8 * 4 * sizeof(WaveRec));
  (void)hipMalloc(&sink, 64);
  (void)hipMalloc(&d_start, 64);
  if (argc > 1 && !strcmp(argv[1], "pmc")) {
    // a short run for rocprofv3 --pmc: one kernel per issue class at 4 waves per SIMD, so that SQ_ACTIVE_INST_VALU /
    // SQ_INSTS_VALU can be read per kernel (does the counter see the half-rate instructions as twice as long?)
    PMC_RUNS
    return 0;
  }
  if (argc > 2 && !strcmp(argv[1], "only")) g_only = argv[2];
  const int Ws[] = {1, 2, 4, 8};
  std::vector<double> table((size_t)N_OPS * 4, 0.0);
  Sweep<0>::go(Ws, 4, table.data());
  printf("\n# summary: cycles per wave-instruction per SIMD\n# %-60s %8s %8s %8s %8s\n", "instruction", "W=1", "W=2", "W=4", "W=8");
  for (int o = 0; o < N_OPS; o++)
    if (!g_only || strstr(kNames[o], g_only)) printf("| %-60s | %6.2f | %6.2f | %6.2f | %6.2f |\n", kNames[o], table[o * 4], table[o * 4 + 1], table[o * 4 + 2], table[o * 4 + 3]);
  if (g_only) return 0;
  printf("\n# EXEC mask: does a partly empty wave64 instruction issue faster?\n");
''')
idx = {o[0]: i for i, o in enumerate(ops)}
for W in (1, 2, 4):
    for lanes in (64, 32, 33, 16, 1):
        for name in ("v_fma_f32", "v_min_f32", "v_cvt_f32_ubyte0", "node-step mix: 2 cvt_ubyte, pk_fma, max3, 2 min, cmp, cndmask"):
            out.append(f"  run<{idx[name]}>({W}, {lanes});\n")
out.append("  return 0;\n}\n")
src = "".join(out)
pmc = "".join(f"run<{idx[n]}>(4, 64);\n    " for n in (
    "v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fmac_f32", "v_min_f32", "v_max_f32", "v_max3_f32", "v_min3_f32", "v_cvt_f32_ubyte0",
    "v_cvt_f32_u32", "v_cmp_le_f32", "v_cmp_eq_u32", "v_cndmask_b32_e64", "v_pk_fma_f32", "v_pk_mul_f32", "v_and_b32", "v_or_b32", "v_add_u32",
    "v_lshlrev_b32", "v_lshrrev_b32", "v_mov_b32", "v_mul_lo_u32", "v_mad_u32_u24", "v_lshl_add_u32", "v_and_or_b32", "v_bfe_u32", "v_rcp_f32",
    "v_sqrt_f32", "v_mad_u64_u32", "v_fma_f64x", "v_readfirstlane_b32",
    "node-step mix: 2 cvt_ubyte, pk_fma, max3, 2 min, cmp, cndmask"))
src = src.replace("PMC_RUNS", pmc)
import os
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_issue.hip"), "w").write(src)
