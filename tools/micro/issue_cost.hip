// Issue cost of the instruction classes of the shift-reuse loop, per wavefront, at 1 / 2 / 3 waves
// per SIMD (round 3).  Every body is an unrolled block of instructions without data dependences
// between consecutive instructions; the loop overhead (s_sub + s_cbranch per block) is measured by
// the empty body.  Reported: ns per block per wave, and the same in cycles of a 2.1 GHz clock.
// build: hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define R4(x) x x x x
#define R8(x) R4(x) R4(x)

// registers v[40..103] are scratch for the bodies; s[40..71] too
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55", \
             "v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71", \
             "v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87", \
             "v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103", \
             "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55", \
             "s56","s57","s58","s59","s60","s61","s62","s63","vcc","scc","m0","memory"

// 32 independent adds into v[40..103] pairs from v[104:105]... use %1 as the addend
#define ADD4(base) "v_add_f64 v[" #base ":" #base "+1], v[96:97], v[" #base ":" #base "+1]\n\t"
static __device__ __forceinline__ void body_empty() {}

template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, const char *tab, double *out) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1.0 + i;
    __syncthreads();
    unsigned addr = (threadIdx.x & 63) * 16;
    unsigned long long tp = (unsigned long long)tab;
    unsigned tlo = (unsigned)tp, thi = (unsigned)(tp >> 32);
    int n = iters;
    // 32 adds: acc pairs v40..v103 step 2 (32 pairs), operand v[104:105] is not clobbered -> use v[40:41] itself
#define ADDS32 \
    "v_add_f64 v[40:41], v[40:41], v[40:41]\n\tv_add_f64 v[42:43], v[42:43], v[42:43]\n\tv_add_f64 v[44:45], v[44:45], v[44:45]\n\tv_add_f64 v[46:47], v[46:47], v[46:47]\n\t" \
    "v_add_f64 v[48:49], v[48:49], v[48:49]\n\tv_add_f64 v[50:51], v[50:51], v[50:51]\n\tv_add_f64 v[52:53], v[52:53], v[52:53]\n\tv_add_f64 v[54:55], v[54:55], v[54:55]\n\t" \
    "v_add_f64 v[56:57], v[56:57], v[56:57]\n\tv_add_f64 v[58:59], v[58:59], v[58:59]\n\tv_add_f64 v[60:61], v[60:61], v[60:61]\n\tv_add_f64 v[62:63], v[62:63], v[62:63]\n\t" \
    "v_add_f64 v[64:65], v[64:65], v[64:65]\n\tv_add_f64 v[66:67], v[66:67], v[66:67]\n\tv_add_f64 v[68:69], v[68:69], v[68:69]\n\tv_add_f64 v[70:71], v[70:71], v[70:71]\n\t" \
    "v_add_f64 v[72:73], v[72:73], v[72:73]\n\tv_add_f64 v[74:75], v[74:75], v[74:75]\n\tv_add_f64 v[76:77], v[76:77], v[76:77]\n\tv_add_f64 v[78:79], v[78:79], v[78:79]\n\t" \
    "v_add_f64 v[80:81], v[80:81], v[80:81]\n\tv_add_f64 v[82:83], v[82:83], v[82:83]\n\tv_add_f64 v[84:85], v[84:85], v[84:85]\n\tv_add_f64 v[86:87], v[86:87], v[86:87]\n\t" \
    "v_add_f64 v[88:89], v[88:89], v[88:89]\n\tv_add_f64 v[90:91], v[90:91], v[90:91]\n\tv_add_f64 v[92:93], v[92:93], v[92:93]\n\tv_add_f64 v[94:95], v[94:95], v[94:95]\n\t" \
    "v_add_f64 v[96:97], v[96:97], v[96:97]\n\tv_add_f64 v[98:99], v[98:99], v[98:99]\n\tv_add_f64 v[100:101], v[100:101], v[100:101]\n\tv_add_f64 v[102:103], v[102:103], v[102:103]\n\t"
    // 8 x (X + 4 adds)
#define G4(X, a, b, c, d) X "v_add_f64 v[" #a ":" #a "+1], v[" #a ":" #a "+1], v[" #a ":" #a "+1]\n\tv_add_f64 v[" #b ":" #b "+1], v[" #b ":" #b "+1], v[" #b ":" #b "+1]\n\t" \
                            "v_add_f64 v[" #c ":" #c "+1], v[" #c ":" #c "+1], v[" #c ":" #c "+1]\n\tv_add_f64 v[" #d ":" #d "+1], v[" #d ":" #d "+1], v[" #d ":" #d "+1]\n\t"
#define GROUPS8(X) G4(X, 40, 42, 44, 46) G4(X, 48, 50, 52, 54) G4(X, 56, 58, 60, 62) G4(X, 64, 66, 68, 70) \
                   G4(X, 72, 74, 76, 78) G4(X, 80, 82, 84, 86) G4(X, 88, 90, 92, 94) G4(X, 96, 98, 100, 102)
#define READS12 \
    "ds_read_b128 v[40:43], %1 offset:0\n\tds_read_b128 v[44:47], %1 offset:16384\n\tds_read_b128 v[48:51], %1 offset:16\n\tds_read_b128 v[52:55], %1 offset:16400\n\t" \
    "ds_read_b128 v[56:59], %1 offset:32\n\tds_read_b128 v[60:63], %1 offset:16416\n\tds_read_b128 v[64:67], %1 offset:48\n\tds_read_b128 v[68:71], %1 offset:16432\n\t" \
    "ds_read_b128 v[72:75], %1 offset:64\n\tds_read_b128 v[76:79], %1 offset:16448\n\tds_read_b128 v[80:83], %1 offset:80\n\tds_read_b128 v[84:87], %1 offset:16464\n\t"
#define READS12B \
    "ds_read_b128 v[104:107], %1 offset:0\n\tds_read_b128 v[108:111], %1 offset:16384\n\tds_read_b128 v[112:115], %1 offset:16\n\tds_read_b128 v[116:119], %1 offset:16400\n\t" \
    "ds_read_b128 v[120:123], %1 offset:32\n\tds_read_b128 v[124:127], %1 offset:16416\n\tds_read_b128 v[128:131], %1 offset:48\n\tds_read_b128 v[132:135], %1 offset:16432\n\t" \
    "ds_read_b128 v[136:139], %1 offset:64\n\tds_read_b128 v[140:143], %1 offset:16448\n\tds_read_b128 v[144:147], %1 offset:80\n\tds_read_b128 v[148:151], %1 offset:16464\n\t"
#define LOOP(BODY) asm volatile("Lk_%=:\n\t" BODY "s_sub_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 Lk_%=\n\ts_waitcnt lgkmcnt(0)\n\ts_set_gpr_idx_off" \
                                : "+s"(n) : "v"(addr), "s"(tlo), "s"(thi) : CLOB, "v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151")
    if (MODE == 0) LOOP("");
    if (MODE == 1) LOOP(ADDS32);
    if (MODE == 2) LOOP(GROUPS8("s_set_gpr_idx_on s40, 1\n\t"));
    if (MODE == 3) LOOP(GROUPS8("s_nop 0\n\t"));
    if (MODE == 4) LOOP(GROUPS8("s_mov_b32 s41, s42\n\t"));
    if (MODE == 5) LOOP(READS12 "s_waitcnt lgkmcnt(0)\n\t");
    if (MODE == 6) LOOP(READS12B ADDS32 "s_waitcnt lgkmcnt(0)\n\t");          // reads beside independent adds
    if (MODE == 7) LOOP(R8("s_cmp_le_u32 %0, 0\n\ts_cbranch_scc1 Lk_%=\n\t"));   // 8 not-taken compare+branch pairs
    if (MODE == 8) LOOP(R8("s_mov_b32 s40, %2\n\ts_mov_b32 s41, %3\n\t"));       // 16 plain SALU
    if (MODE == 9) LOOP("s_mov_b32 s40, %2\n\ts_mov_b32 s41, %3\n\ts_load_dwordx16 s[44:59], s[40:41], 0\n\ts_waitcnt lgkmcnt(0)\n\t");
    if (MODE == 10) LOOP("s_mov_b32 s40, %2\n\ts_mov_b32 s41, %3\n\ts_load_dwordx16 s[44:59], s[40:41], 0\n\t" ADDS32 "s_waitcnt lgkmcnt(0)\n\t");
    if (MODE == 11) LOOP(READS12B GROUPS8("s_set_gpr_idx_on s40, 1\n\t") "s_waitcnt lgkmcnt(0)\n\t");   // a row: reads + 8 x (idx + 4 adds)
    if (MODE == 12) LOOP(R8("v_fma_f64 v[40:41], v[40:41], v[42:43], v[44:45]\n\tv_fma_f64 v[46:47], v[46:47], v[48:49], v[50:51]\n\tv_fma_f64 v[52:53], v[52:53], v[54:55], v[56:57]\n\tv_fma_f64 v[58:59], v[58:59], v[60:61], v[62:63]\n\t"));  // 32 fma, 4 chains
    if (MODE == 13) LOOP(R8("v_add_f64 v[40:41], v[40:41], v[42:43]\n\tv_add_f64 v[40:41], v[40:41], v[42:43]\n\tv_add_f64 v[40:41], v[40:41], v[42:43]\n\tv_add_f64 v[40:41], v[40:41], v[42:43]\n\t"));  // 32 dependent adds
    if (threadIdx.x == 1000) out[0] = (double)n;
}

int main() {
    char *tab; double *out;
    CK(hipMalloc(&tab, 4096)); CK(hipMemset(tab, 0, 4096)); CK(hipMalloc(&out, 64));
    const char *names[] = {"empty loop", "32 v_add_f64 (independent)", "8 x (s_set_gpr_idx_on + 4 v_add_f64)",
                           "8 x (s_nop + 4 v_add_f64)", "8 x (s_mov + 4 v_add_f64)", "12 ds_read_b128 + wait",
                           "12 ds_read_b128 + 32 v_add_f64 + wait", "8 x (s_cmp + s_cbranch not taken)", "16 s_mov",
                           "s_load_dwordx16 + wait", "s_load_dwordx16 + 32 v_add_f64 + wait",
                           "row: 12 ds_read_b128 + 8 x (idx + 4 adds) + wait", "32 v_fma_f64 in 4 chains",
                           "32 dependent v_add_f64"};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int wps = 1; wps <= 3; ++wps) {
        printf("---- %d wave(s) per SIMD (%d workgroups of 256 threads, 1 CU each ~)\n", wps, 256 * wps);
        for (int mode = 0; mode < 14; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
#define L(M) case M: hipLaunchKernelGGL(k<M>, dim3(256 * wps), dim3(256), 49152, 0, iters, tab, out); break;
                switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) best = ms < best ? ms : best;
            }
            const double ns = best * 1e6 / iters;
            printf("%-52s %8.1f ns per block per wave = %6.0f clk at 2.1 GHz | per SIMD %6.0f clk\n", names[mode], ns, ns * 2.1, ns * 2.1 / wps);
        }
    }
    CK(hipGetLastError());
    return 0;
}
