// Issue-rate probe for the instructions of the DP sweep (gfx950): how many cycles one SIMD spends
// per wave-instruction when all its wave slots are full (8 waves per SIMD, every CU busy), so that
// the sweep's instruction stream can be priced in cycles instead of instruction counts.
//   hipcc --offload-arch=gfx950 -O3 -o valu_tp valu_tp.hip && ./valu_tp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITERS 4096
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// eight independent instructions per loop trip; operands live in v[10:41], results in v[42:57]
#define PROBE(name, body)                                                             \
__global__ void __launch_bounds__(256) name(float* out)                               \
{                                                                                     \
    asm volatile(                                                                     \
        "v_mov_b32 v10, 1.0\n v_mov_b32 v11, 0.5\n v_mov_b32 v12, 2.0\n v_mov_b32 v13, 0.5\n" \
        "v_mov_b32 v14, 1.0\n v_mov_b32 v15, 0.5\n v_mov_b32 v16, 2.0\n v_mov_b32 v17, 0.5\n" \
        "v_mov_b32 v18, 0\n v_mov_b32 v19, 0x3ff00000\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0x3fe00000\n" \
        "v_mov_b32 v22, 0\n v_mov_b32 v23, 0x40000000\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0x3fe00000\n" \
        "v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n"          \
        "s_mov_b32 s20, %0\n s_mov_b64 s[4:5], exec\n s_mov_b64 s[6:7], 0\n s_mov_b64 s[8:9], exec\n s_mov_b64 s[10:11], 0\n s_mov_b64 vcc, exec\n"                                                        \
        "1:\n" body body body body                                                                    \
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"          \
        :: "n"(ITERS)                                                                 \
        : "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25", \
          "v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41", \
          "v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57", \
          "s4","s5","s6","s7","s8","s9","s10","s11","s20","vcc","scc","memory");                                                \
    if (out == nullptr) out[threadIdx.x] = 0.0f;                                      \
}

#define R8(i) i i i i i i i i
PROBE(p_fma_f32,    "v_fma_f32 v42, v10, v11, v12\n v_fma_f32 v43, v11, v12, v13\n v_fma_f32 v44, v12, v13, v14\n v_fma_f32 v45, v13, v14, v15\n"
                    "v_fma_f32 v46, v14, v15, v16\n v_fma_f32 v47, v15, v16, v17\n v_fma_f32 v48, v16, v17, v10\n v_fma_f32 v49, v17, v10, v11\n")
PROBE(p_add_f32,    "v_add_f32 v42, v10, v11\n v_add_f32 v43, v11, v12\n v_add_f32 v44, v12, v13\n v_add_f32 v45, v13, v14\n"
                    "v_add_f32 v46, v14, v15\n v_add_f32 v47, v15, v16\n v_add_f32 v48, v16, v17\n v_add_f32 v49, v17, v10\n")
PROBE(p_pk_fma_f32, "v_pk_fma_f32 v[42:43], v[10:11], v[12:13], v[14:15]\n v_pk_fma_f32 v[44:45], v[12:13], v[14:15], v[16:17]\n"
                    "v_pk_fma_f32 v[46:47], v[14:15], v[16:17], v[10:11]\n v_pk_fma_f32 v[48:49], v[16:17], v[10:11], v[12:13]\n"
                    "v_pk_fma_f32 v[50:51], v[10:11], v[12:13], v[14:15]\n v_pk_fma_f32 v[52:53], v[12:13], v[14:15], v[16:17]\n"
                    "v_pk_fma_f32 v[54:55], v[14:15], v[16:17], v[10:11]\n v_pk_fma_f32 v[56:57], v[16:17], v[10:11], v[12:13]\n")
PROBE(p_pk_add_f32, "v_pk_add_f32 v[42:43], v[10:11], v[12:13]\n v_pk_add_f32 v[44:45], v[12:13], v[14:15]\n"
                    "v_pk_add_f32 v[46:47], v[14:15], v[16:17]\n v_pk_add_f32 v[48:49], v[16:17], v[10:11]\n"
                    "v_pk_add_f32 v[50:51], v[10:11], v[12:13]\n v_pk_add_f32 v[52:53], v[12:13], v[14:15]\n"
                    "v_pk_add_f32 v[54:55], v[14:15], v[16:17]\n v_pk_add_f32 v[56:57], v[16:17], v[10:11]\n")
PROBE(p_pk_mul_f32, "v_pk_mul_f32 v[42:43], v[10:11], v[12:13]\n v_pk_mul_f32 v[44:45], v[12:13], v[14:15]\n"
                    "v_pk_mul_f32 v[46:47], v[14:15], v[16:17]\n v_pk_mul_f32 v[48:49], v[16:17], v[10:11]\n"
                    "v_pk_mul_f32 v[50:51], v[10:11], v[12:13]\n v_pk_mul_f32 v[52:53], v[12:13], v[14:15]\n"
                    "v_pk_mul_f32 v[54:55], v[14:15], v[16:17]\n v_pk_mul_f32 v[56:57], v[16:17], v[10:11]\n")
PROBE(p_add_f64,    "v_add_f64 v[42:43], v[18:19], v[20:21]\n v_add_f64 v[44:45], v[20:21], v[22:23]\n"
                    "v_add_f64 v[46:47], v[22:23], v[24:25]\n v_add_f64 v[48:49], v[24:25], v[18:19]\n"
                    "v_add_f64 v[50:51], v[18:19], v[20:21]\n v_add_f64 v[52:53], v[20:21], v[22:23]\n"
                    "v_add_f64 v[54:55], v[22:23], v[24:25]\n v_add_f64 v[56:57], v[24:25], v[18:19]\n")
PROBE(p_mul_f64,    "v_mul_f64 v[42:43], v[18:19], v[20:21]\n v_mul_f64 v[44:45], v[20:21], v[22:23]\n"
                    "v_mul_f64 v[46:47], v[22:23], v[24:25]\n v_mul_f64 v[48:49], v[24:25], v[18:19]\n"
                    "v_mul_f64 v[50:51], v[18:19], v[20:21]\n v_mul_f64 v[52:53], v[20:21], v[22:23]\n"
                    "v_mul_f64 v[54:55], v[22:23], v[24:25]\n v_mul_f64 v[56:57], v[24:25], v[18:19]\n")
PROBE(p_max_f64,    "v_max_f64 v[42:43], v[18:19], v[20:21]\n v_max_f64 v[44:45], v[20:21], v[22:23]\n"
                    "v_max_f64 v[46:47], v[22:23], v[24:25]\n v_max_f64 v[48:49], v[24:25], v[18:19]\n"
                    "v_max_f64 v[50:51], v[18:19], v[20:21]\n v_max_f64 v[52:53], v[20:21], v[22:23]\n"
                    "v_max_f64 v[54:55], v[22:23], v[24:25]\n v_max_f64 v[56:57], v[24:25], v[18:19]\n")
PROBE(p_cvt_f64_f32, "v_cvt_f64_f32 v[42:43], v10\n v_cvt_f64_f32 v[44:45], v11\n v_cvt_f64_f32 v[46:47], v12\n v_cvt_f64_f32 v[48:49], v13\n"
                     "v_cvt_f64_f32 v[50:51], v14\n v_cvt_f64_f32 v[52:53], v15\n v_cvt_f64_f32 v[54:55], v16\n v_cvt_f64_f32 v[56:57], v17\n")
PROBE(p_cvt_f32_f64, "v_cvt_f32_f64 v42, v[18:19]\n v_cvt_f32_f64 v43, v[20:21]\n v_cvt_f32_f64 v44, v[22:23]\n v_cvt_f32_f64 v45, v[24:25]\n"
                     "v_cvt_f32_f64 v46, v[18:19]\n v_cvt_f32_f64 v47, v[20:21]\n v_cvt_f32_f64 v48, v[22:23]\n v_cvt_f32_f64 v49, v[24:25]\n")
PROBE(p_cmp_lt_u64,  R8("v_cmp_lt_u64 vcc, v[18:19], v[20:21]\n"))
PROBE(p_cmp_lt_f32,  R8("v_cmp_lt_f32 vcc, v10, v11\n"))
PROBE(p_cndmask,     "v_cndmask_b32 v42, v10, v11, vcc\n v_cndmask_b32 v43, v11, v12, vcc\n v_cndmask_b32 v44, v12, v13, vcc\n v_cndmask_b32 v45, v13, v14, vcc\n"
                     "v_cndmask_b32 v46, v14, v15, vcc\n v_cndmask_b32 v47, v15, v16, vcc\n v_cndmask_b32 v48, v16, v17, vcc\n v_cndmask_b32 v49, v17, v10, vcc\n")
PROBE(p_floor_f32,   "v_floor_f32 v42, v10\n v_floor_f32 v43, v11\n v_floor_f32 v44, v12\n v_floor_f32 v45, v13\n"
                     "v_floor_f32 v46, v14\n v_floor_f32 v47, v15\n v_floor_f32 v48, v16\n v_floor_f32 v49, v17\n")
PROBE(p_rndne_f32,   "v_rndne_f32 v42, v10\n v_rndne_f32 v43, v11\n v_rndne_f32 v44, v12\n v_rndne_f32 v45, v13\n"
                     "v_rndne_f32 v46, v14\n v_rndne_f32 v47, v15\n v_rndne_f32 v48, v16\n v_rndne_f32 v49, v17\n")
PROBE(p_cvt_i32_f32, "v_cvt_i32_f32 v42, v10\n v_cvt_i32_f32 v43, v11\n v_cvt_i32_f32 v44, v12\n v_cvt_i32_f32 v45, v13\n"
                     "v_cvt_i32_f32 v46, v14\n v_cvt_i32_f32 v47, v15\n v_cvt_i32_f32 v48, v16\n v_cvt_i32_f32 v49, v17\n")
PROBE(p_mov_b32,     "v_mov_b32 v42, v10\n v_mov_b32 v43, v11\n v_mov_b32 v44, v12\n v_mov_b32 v45, v13\n"
                     "v_mov_b32 v46, v14\n v_mov_b32 v47, v15\n v_mov_b32 v48, v16\n v_mov_b32 v49, v17\n")
PROBE(p_mov_b64,     "v_mov_b64 v[42:43], v[10:11]\n v_mov_b64 v[44:45], v[12:13]\n v_mov_b64 v[46:47], v[14:15]\n v_mov_b64 v[48:49], v[16:17]\n"
                     "v_mov_b64 v[50:51], v[10:11]\n v_mov_b64 v[52:53], v[12:13]\n v_mov_b64 v[54:55], v[14:15]\n v_mov_b64 v[56:57], v[16:17]\n")
PROBE(p_pk_mov_b32,  "v_pk_mov_b32 v[42:43], v[10:11], v[12:13] op_sel:[0,1]\n v_pk_mov_b32 v[44:45], v[12:13], v[14:15] op_sel:[0,1]\n"
                     "v_pk_mov_b32 v[46:47], v[14:15], v[16:17] op_sel:[0,1]\n v_pk_mov_b32 v[48:49], v[16:17], v[10:11] op_sel:[0,1]\n"
                     "v_pk_mov_b32 v[50:51], v[10:11], v[12:13] op_sel:[0,1]\n v_pk_mov_b32 v[52:53], v[12:13], v[14:15] op_sel:[0,1]\n"
                     "v_pk_mov_b32 v[54:55], v[14:15], v[16:17] op_sel:[0,1]\n v_pk_mov_b32 v[56:57], v[16:17], v[10:11] op_sel:[0,1]\n")
PROBE(p_lshl_add,    "v_lshl_add_u32 v42, v26, 4, v27\n v_lshl_add_u32 v43, v27, 4, v28\n v_lshl_add_u32 v44, v28, 4, v29\n v_lshl_add_u32 v45, v29, 4, v26\n"
                     "v_lshl_add_u32 v46, v26, 4, v27\n v_lshl_add_u32 v47, v27, 4, v28\n v_lshl_add_u32 v48, v28, 4, v29\n v_lshl_add_u32 v49, v29, 4, v26\n")
PROBE(p_max_f32,     "v_max_f32 v42, v10, v11\n v_max_f32 v43, v11, v12\n v_max_f32 v44, v12, v13\n v_max_f32 v45, v13, v14\n"
                     "v_max_f32 v46, v14, v15\n v_max_f32 v47, v15, v16\n v_max_f32 v48, v16, v17\n v_max_f32 v49, v17, v10\n")
PROBE(p_fma_f64,     "v_fma_f64 v[42:43], v[18:19], v[20:21], v[22:23]\n v_fma_f64 v[44:45], v[20:21], v[22:23], v[24:25]\n"
                     "v_fma_f64 v[46:47], v[22:23], v[24:25], v[18:19]\n v_fma_f64 v[48:49], v[24:25], v[18:19], v[20:21]\n"
                     "v_fma_f64 v[50:51], v[18:19], v[20:21], v[22:23]\n v_fma_f64 v[52:53], v[20:21], v[22:23], v[24:25]\n"
                     "v_fma_f64 v[54:55], v[22:23], v[24:25], v[18:19]\n v_fma_f64 v[56:57], v[24:25], v[18:19], v[20:21]\n")
PROBE(p_s_nop,       R8("s_nop 0\n"))

PROBE(p_mul_f32,     "v_mul_f32 v42, v10, v11\n v_mul_f32 v43, v11, v12\n v_mul_f32 v44, v12, v13\n v_mul_f32 v45, v13, v14\n"
                     "v_mul_f32 v46, v14, v15\n v_mul_f32 v47, v15, v16\n v_mul_f32 v48, v16, v17\n v_mul_f32 v49, v17, v10\n")
PROBE(p_cndmask_e64, "v_cndmask_b32 v42, v10, v11, s[4:5]\n v_cndmask_b32 v43, v11, v12, s[4:5]\n v_cndmask_b32 v44, v12, v13, s[4:5]\n v_cndmask_b32 v45, v13, v14, s[4:5]\n"
                     "v_cndmask_b32 v46, v14, v15, s[4:5]\n v_cndmask_b32 v47, v15, v16, s[4:5]\n v_cndmask_b32 v48, v16, v17, s[4:5]\n v_cndmask_b32 v49, v17, v10, s[4:5]\n")
PROBE(p_cndmask_wr,  "v_cndmask_b32 v42, v10, v11, vcc\n v_cndmask_b32 v43, v11, v42, vcc\n v_cndmask_b32 v44, v12, v43, vcc\n v_cndmask_b32 v45, v13, v44, vcc\n"
                     "v_cndmask_b32 v46, v14, v45, vcc\n v_cndmask_b32 v47, v15, v46, vcc\n v_cndmask_b32 v48, v16, v47, vcc\n v_cndmask_b32 v49, v17, v48, vcc\n")
PROBE(p_cmp_cnd,     "v_cmp_lt_f32 vcc, v10, v11\n v_cndmask_b32 v42, v10, v11, vcc\n v_cmp_lt_f32 vcc, v12, v13\n v_cndmask_b32 v43, v12, v13, vcc\n"
                     "v_cmp_lt_f32 vcc, v14, v15\n v_cndmask_b32 v44, v14, v15, vcc\n v_cmp_lt_f32 vcc, v16, v17\n v_cndmask_b32 v45, v16, v17, vcc\n")
PROBE(p_cmp_cnd3,    "v_cmp_lt_f32 vcc, v10, v11\n v_cndmask_b32 v42, v10, v11, vcc\n v_cndmask_b32 v43, v12, v13, vcc\n v_cndmask_b32 v44, v14, v15, vcc\n"
                     "v_cmp_lt_f32 vcc, v14, v15\n v_cndmask_b32 v45, v14, v15, vcc\n v_cndmask_b32 v46, v16, v17, vcc\n v_cndmask_b32 v47, v10, v17, vcc\n")
PROBE(p_cmp64_cnd,   "v_cmp_lt_f32 s[4:5], v10, v11\n v_cndmask_b32 v42, v10, v11, s[4:5]\n v_cmp_lt_f32 s[6:7], v12, v13\n v_cndmask_b32 v43, v12, v13, s[6:7]\n"
                     "v_cmp_lt_f32 s[8:9], v14, v15\n v_cndmask_b32 v44, v14, v15, s[8:9]\n v_cmp_lt_f32 s[10:11], v16, v17\n v_cndmask_b32 v45, v16, v17, s[10:11]\n")
PROBE(p_bfi_b32,     "v_bfi_b32 v42, v26, v10, v11\n v_bfi_b32 v43, v27, v11, v12\n v_bfi_b32 v44, v28, v12, v13\n v_bfi_b32 v45, v29, v13, v14\n"
                     "v_bfi_b32 v46, v26, v14, v15\n v_bfi_b32 v47, v27, v15, v16\n v_bfi_b32 v48, v28, v16, v17\n v_bfi_b32 v49, v29, v17, v10\n")
PROBE(p_min_u32,     "v_min_u32 v42, v10, v11\n v_min_u32 v43, v11, v12\n v_min_u32 v44, v12, v13\n v_min_u32 v45, v13, v14\n"
                     "v_min_u32 v46, v14, v15\n v_min_u32 v47, v15, v16\n v_min_u32 v48, v16, v17\n v_min_u32 v49, v17, v10\n")
PROBE(p_sub_u32,     "v_sub_u32 v42, v10, v11\n v_sub_u32 v43, v11, v12\n v_sub_u32 v44, v12, v13\n v_sub_u32 v45, v13, v14\n"
                     "v_sub_u32 v46, v14, v15\n v_sub_u32 v47, v15, v16\n v_sub_u32 v48, v16, v17\n v_sub_u32 v49, v17, v10\n")
PROBE(p_ashrrev,     "v_ashrrev_i32 v42, 31, v10\n v_ashrrev_i32 v43, 31, v11\n v_ashrrev_i32 v44, 31, v12\n v_ashrrev_i32 v45, 31, v13\n"
                     "v_ashrrev_i32 v46, 31, v14\n v_ashrrev_i32 v47, 31, v15\n v_ashrrev_i32 v48, 31, v16\n v_ashrrev_i32 v49, 31, v17\n")
PROBE(p_med3_f32,    "v_med3_f32 v42, v10, v11, v12\n v_med3_f32 v43, v11, v12, v13\n v_med3_f32 v44, v12, v13, v14\n v_med3_f32 v45, v13, v14, v15\n"
                     "v_med3_f32 v46, v14, v15, v16\n v_med3_f32 v47, v15, v16, v17\n v_med3_f32 v48, v16, v17, v10\n v_med3_f32 v49, v17, v10, v11\n")
PROBE(p_fract_f32,   "v_fract_f32 v42, v10\n v_fract_f32 v43, v11\n v_fract_f32 v44, v12\n v_fract_f32 v45, v13\n"
                     "v_fract_f32 v46, v14\n v_fract_f32 v47, v15\n v_fract_f32 v48, v16\n v_fract_f32 v49, v17\n")
PROBE(p_fmac_f32,    "v_fmac_f32 v42, v10, v11\n v_fmac_f32 v43, v11, v12\n v_fmac_f32 v44, v12, v13\n v_fmac_f32 v45, v13, v14\n"
                     "v_fmac_f32 v46, v14, v15\n v_fmac_f32 v47, v15, v16\n v_fmac_f32 v48, v16, v17\n v_fmac_f32 v49, v17, v10\n")
PROBE(p_add_f32_sgpr, "v_add_f32 v42, s4, v11\n v_add_f32 v43, s5, v12\n v_add_f32 v44, s6, v13\n v_add_f32 v45, s7, v14\n"
                     "v_add_f32 v46, s8, v15\n v_add_f32 v47, s9, v16\n v_add_f32 v48, s10, v17\n v_add_f32 v49, s11, v10\n")
PROBE(p_max_f32_sgpr, "v_max_f32 v42, s4, v11\n v_max_f32 v43, s5, v12\n v_max_f32 v44, s6, v13\n v_max_f32 v45, s7, v14\n"
                     "v_max_f32 v46, s8, v15\n v_max_f32 v47, s9, v16\n v_max_f32 v48, s10, v17\n v_max_f32 v49, s11, v10\n")
PROBE(p_pk_fma_sgpr, "v_pk_fma_f32 v[42:43], v[10:11], s[4:5], v[14:15]\n v_pk_fma_f32 v[44:45], v[12:13], s[6:7], v[16:17]\n"
                    "v_pk_fma_f32 v[46:47], v[14:15], s[8:9], v[10:11]\n v_pk_fma_f32 v[48:49], v[16:17], s[10:11], v[12:13]\n"
                    "v_pk_fma_f32 v[50:51], v[10:11], s[4:5], v[14:15]\n v_pk_fma_f32 v[52:53], v[12:13], s[6:7], v[16:17]\n"
                    "v_pk_fma_f32 v[54:55], v[14:15], s[8:9], v[10:11]\n v_pk_fma_f32 v[56:57], v[16:17], s[10:11], v[12:13]\n")
PROBE(p_mix_fma_max, "v_fma_f32 v42, v10, v11, v12\n v_max_f32 v43, v11, v12\n v_fma_f32 v44, v12, v13, v14\n v_max_f32 v45, v13, v14\n"
                    "v_fma_f32 v46, v14, v15, v16\n v_max_f32 v47, v15, v16\n v_fma_f32 v48, v16, v17, v10\n v_max_f32 v49, v17, v10\n")
PROBE(p_cmp_lt_u32,  R8("v_cmp_lt_u32 vcc, v10, v11\n"))
PROBE(p_cmp_class,   R8("v_cmp_lt_f32 s[6:7], v10, v11\n"))


// two-pipe hypothesis: do "fast" (fma/add/mul, VGPR operands) and "slow" (max, cvt, f64, pk, ...) instructions overlap?
PROBE(p_mix_6fma_2max, "v_fma_f32 v42, v10, v11, v12\n v_fma_f32 v43, v11, v12, v13\n v_fma_f32 v44, v12, v13, v14\n v_max_f32 v45, v13, v14\n"
                       "v_fma_f32 v46, v14, v15, v16\n v_fma_f32 v47, v15, v16, v17\n v_fma_f32 v48, v16, v17, v10\n v_max_f32 v49, v17, v10\n")
PROBE(p_mix_2fma_6max, "v_max_f32 v42, v10, v11\n v_max_f32 v43, v11, v12\n v_max_f32 v44, v12, v13\n v_fma_f32 v45, v13, v14, v15\n"
                       "v_max_f32 v46, v14, v15\n v_max_f32 v47, v15, v16\n v_max_f32 v48, v16, v17\n v_fma_f32 v49, v17, v10, v11\n")
PROBE(p_mix_fma_f64,   "v_fma_f32 v42, v10, v11, v12\n v_add_f64 v[50:51], v[18:19], v[20:21]\n v_fma_f32 v44, v12, v13, v14\n v_add_f64 v[52:53], v[20:21], v[22:23]\n"
                       "v_fma_f32 v46, v14, v15, v16\n v_add_f64 v[54:55], v[22:23], v[24:25]\n v_fma_f32 v48, v16, v17, v10\n v_add_f64 v[56:57], v[24:25], v[18:19]\n")
PROBE(p_mix_fma_cvt,   "v_fma_f32 v42, v10, v11, v12\n v_cvt_f64_f32 v[50:51], v10\n v_fma_f32 v44, v12, v13, v14\n v_cvt_i32_f32 v52, v11\n"
                       "v_fma_f32 v46, v14, v15, v16\n v_floor_f32 v53, v12\n v_fma_f32 v48, v16, v17, v10\n v_cvt_f32_f64 v54, v[18:19]\n")
PROBE(p_mix_fma_pk,    "v_fma_f32 v42, v10, v11, v12\n v_pk_fma_f32 v[50:51], v[10:11], v[12:13], v[14:15]\n v_fma_f32 v44, v12, v13, v14\n v_pk_add_f32 v[52:53], v[12:13], v[14:15]\n"
                       "v_fma_f32 v46, v14, v15, v16\n v_pk_mul_f32 v[54:55], v[14:15], v[16:17]\n v_fma_f32 v48, v16, v17, v10\n v_pk_fma_f32 v[56:57], v[16:17], v[10:11], v[12:13]\n")
PROBE(p_mix_fma_cmpcnd, "v_fma_f32 v42, v10, v11, v12\n v_cmp_lt_f32 vcc, v10, v11\n v_fma_f32 v44, v12, v13, v14\n v_cndmask_b32 v50, v10, v11, vcc\n"
                       "v_fma_f32 v46, v14, v15, v16\n v_cmp_lt_f32 vcc, v14, v15\n v_fma_f32 v48, v16, v17, v10\n v_cndmask_b32 v51, v14, v15, vcc\n")
PROBE(p_mix_3fma_1max, "v_fma_f32 v42, v10, v11, v12\n v_fma_f32 v43, v11, v12, v13\n v_fma_f32 v44, v12, v13, v14\n v_max_f32 v45, v13, v14\n"
                       "v_mul_f32 v46, v14, v15\n v_add_f32 v47, v15, v16\n v_fma_f32 v48, v16, v17, v10\n v_cvt_i32_f32 v49, v17\n")
PROBE(p_mix_fma_lds,   "v_lshlrev_b32 v30, 4, v0\n v_fma_f32 v42, v10, v11, v12\n v_fma_f32 v43, v11, v12, v13\n ds_read_b128 v[50:53], v30\n v_fma_f32 v44, v12, v13, v14\n v_fma_f32 v45, v13, v14, v15\n"
                       "v_fma_f32 v46, v14, v15, v16\n v_fma_f32 v47, v15, v16, v17\n v_fma_f32 v48, v16, v17, v10\n v_fma_f32 v49, v17, v10, v11\n s_waitcnt lgkmcnt(0)\n")
// operand kinds on the fast instructions
PROBE(p_mul_f32_sgpr,  "v_mul_f32 v42, s4, v11\n v_mul_f32 v43, s5, v12\n v_mul_f32 v44, s6, v13\n v_mul_f32 v45, s7, v14\n"
                       "v_mul_f32 v46, s8, v15\n v_mul_f32 v47, s9, v16\n v_mul_f32 v48, s10, v17\n v_mul_f32 v49, s11, v10\n")
PROBE(p_fma_f32_sgpr,  "v_fma_f32 v42, v10, s4, v12\n v_fma_f32 v43, v11, s5, v13\n v_fma_f32 v44, v12, s6, v14\n v_fma_f32 v45, v13, s7, v15\n"
                       "v_fma_f32 v46, v14, s8, v16\n v_fma_f32 v47, v15, s9, v17\n v_fma_f32 v48, v16, s10, v10\n v_fma_f32 v49, v17, s11, v11\n")
PROBE(p_add_f32_const, "v_add_f32 v42, 0.5, v11\n v_add_f32 v43, 1.0, v12\n v_add_f32 v44, 2.0, v13\n v_add_f32 v45, 0.5, v14\n"
                       "v_add_f32 v46, 1.0, v15\n v_add_f32 v47, 2.0, v16\n v_add_f32 v48, 0.5, v17\n v_add_f32 v49, 1.0, v10\n")
PROBE(p_add_f32_lit,   "v_add_f32 v42, 0x4b000000, v11\n v_add_f32 v43, 0x4b000000, v12\n v_add_f32 v44, 0x4b000000, v13\n v_add_f32 v45, 0x4b000000, v14\n"
                       "v_add_f32 v46, 0x4b000000, v15\n v_add_f32 v47, 0x4b000000, v16\n v_add_f32 v48, 0x4b000000, v17\n v_add_f32 v49, 0x4b000000, v10\n")
PROBE(p_mov_b32_sgpr,  "v_mov_b32 v42, s4\n v_mov_b32 v43, s5\n v_mov_b32 v44, s6\n v_mov_b32 v45, s7\n"
                       "v_mov_b32 v46, s8\n v_mov_b32 v47, s9\n v_mov_b32 v48, s10\n v_mov_b32 v49, s11\n")
PROBE(p_sub_f32,       "v_sub_f32 v42, v10, v11\n v_sub_f32 v43, v11, v12\n v_sub_f32 v44, v12, v13\n v_sub_f32 v45, v13, v14\n"
                       "v_sub_f32 v46, v14, v15\n v_sub_f32 v47, v15, v16\n v_sub_f32 v48, v16, v17\n v_sub_f32 v49, v17, v10\n")
PROBE(p_add_u32,       "v_add_u32 v42, v26, v27\n v_add_u32 v43, v27, v28\n v_add_u32 v44, v28, v29\n v_add_u32 v45, v29, v26\n"
                       "v_add_u32 v46, v26, v27\n v_add_u32 v47, v27, v28\n v_add_u32 v48, v28, v29\n v_add_u32 v49, v29, v26\n")
PROBE(p_lshlrev,       "v_lshlrev_b32 v42, 4, v26\n v_lshlrev_b32 v43, 4, v27\n v_lshlrev_b32 v44, 4, v28\n v_lshlrev_b32 v45, 4, v29\n"
                       "v_lshlrev_b32 v46, 4, v26\n v_lshlrev_b32 v47, 4, v27\n v_lshlrev_b32 v48, 4, v28\n v_lshlrev_b32 v49, 4, v29\n")
PROBE(p_and_b32,       "v_and_b32 v42, v26, v27\n v_and_b32 v43, v27, v28\n v_and_b32 v44, v28, v29\n v_and_b32 v45, v29, v26\n"
                       "v_and_b32 v46, v26, v27\n v_and_b32 v47, v27, v28\n v_and_b32 v48, v28, v29\n v_and_b32 v49, v29, v26\n")
PROBE(p_fma_f32_neg,   "v_fma_f32 v42, -v10, v11, v12\n v_fma_f32 v43, -v11, v12, v13\n v_fma_f32 v44, -v12, v13, v14\n v_fma_f32 v45, -v13, v14, v15\n"
                       "v_fma_f32 v46, -v14, v15, v16\n v_fma_f32 v47, -v15, v16, v17\n v_fma_f32 v48, -v16, v17, v10\n v_fma_f32 v49, -v17, v10, v11\n")
PROBE(p_add_f32_e64abs, "v_add_f32_e64 v42, |v10|, v11\n v_add_f32_e64 v43, |v11|, v12\n v_add_f32_e64 v44, |v12|, v13\n v_add_f32_e64 v45, |v13|, v14\n"
                       "v_add_f32_e64 v46, |v14|, v15\n v_add_f32_e64 v47, |v15|, v16\n v_add_f32_e64 v48, |v16|, v17\n v_add_f32_e64 v49, |v17|, v10\n")
PROBE(p_cvt_f32_i32,   "v_cvt_f32_i32 v42, v26\n v_cvt_f32_i32 v43, v27\n v_cvt_f32_i32 v44, v28\n v_cvt_f32_i32 v45, v29\n"
                       "v_cvt_f32_i32 v46, v26\n v_cvt_f32_i32 v47, v27\n v_cvt_f32_i32 v48, v28\n v_cvt_f32_i32 v49, v29\n")
PROBE(p_max3_f32,      "v_max3_f32 v42, v10, v11, v12\n v_max3_f32 v43, v11, v12, v13\n v_max3_f32 v44, v12, v13, v14\n v_max3_f32 v45, v13, v14, v15\n"
                       "v_max3_f32 v46, v14, v15, v16\n v_max3_f32 v47, v15, v16, v17\n v_max3_f32 v48, v16, v17, v10\n v_max3_f32 v49, v17, v10, v11\n")

// LDS probes: conflict-free addresses lane*16 (b128 / b96) or lane*4 (b32)
PROBE(p_ds_read_b128, "v_lshlrev_b32 v30, 4, v0\n" R8("ds_read_b128 v[42:45], v30\n") "s_waitcnt lgkmcnt(0)\n")
PROBE(p_ds_read_b96,  "v_lshlrev_b32 v30, 4, v0\n" R8("ds_read_b96 v[42:44], v30\n") "s_waitcnt lgkmcnt(0)\n")
PROBE(p_ds_read2_b32, "v_lshlrev_b32 v30, 2, v0\n" R8("ds_read2_b32 v[42:43], v30 offset1:1\n") "s_waitcnt lgkmcnt(0)\n")
PROBE(p_ds_read_b64,  "v_lshlrev_b32 v30, 3, v0\n" R8("ds_read_b64 v[42:43], v30\n") "s_waitcnt lgkmcnt(0)\n")
PROBE(p_ds_write_b128, "v_lshlrev_b32 v30, 4, v0\n" R8("ds_write_b128 v30, v[10:13]\n") "s_waitcnt lgkmcnt(0)\n")
PROBE(p_ds_write_b96, "v_lshlrev_b32 v30, 4, v0\n" R8("ds_write_b96 v30, v[10:12]\n") "s_waitcnt lgkmcnt(0)\n")

int main(int argc, char** argv)
{
    int waves_per_simd = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd;          // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* dummy; CK(hipMalloc(&dummy, 4096));
    printf("%d CUs, %d waves per SIMD, %d x 32 instructions per wave; clock rate prop %.0f MHz\n", cus, waves_per_simd, ITERS, prop.clockRate / 1e3);
    auto run = [&](const char* name, void (*k)(float*), size_t lds) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, dummy);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) {
                const double inst_per_simd = (double)ITERS * 32 * waves_per_simd;
                printf("%-16s %8.3f ms   %6.2f ns per wave-instruction per SIMD = %5.2f cycles at 2.4 GHz\n", name, ms,
                       ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
            }
        }
    };
#define RUN(k) run(#k, k, 0)
    RUN(p_fma_f32); RUN(p_add_f32); RUN(p_max_f32); RUN(p_pk_fma_f32); RUN(p_pk_add_f32); RUN(p_pk_mul_f32);
    RUN(p_add_f64); RUN(p_mul_f64); RUN(p_fma_f64); RUN(p_max_f64); RUN(p_cvt_f64_f32); RUN(p_cvt_f32_f64);
    RUN(p_cmp_lt_u64); RUN(p_cmp_lt_f32); RUN(p_cndmask); RUN(p_floor_f32); RUN(p_rndne_f32); RUN(p_cvt_i32_f32);
    RUN(p_mul_f32); RUN(p_fmac_f32); RUN(p_cndmask_e64); RUN(p_cndmask_wr); RUN(p_cmp_cnd); RUN(p_cmp_cnd3); RUN(p_cmp64_cnd); RUN(p_bfi_b32); RUN(p_min_u32); RUN(p_sub_u32); RUN(p_ashrrev); RUN(p_med3_f32); RUN(p_fract_f32); RUN(p_add_f32_sgpr); RUN(p_max_f32_sgpr); RUN(p_pk_fma_sgpr); RUN(p_mix_fma_max); RUN(p_cmp_lt_u32); RUN(p_cmp_class);
    RUN(p_mix_6fma_2max); RUN(p_mix_2fma_6max); RUN(p_mix_fma_f64); RUN(p_mix_fma_cvt); RUN(p_mix_fma_pk); RUN(p_mix_fma_cmpcnd); RUN(p_mix_3fma_1max);
    run("p_mix_fma_lds", p_mix_fma_lds, 8192);
    RUN(p_mul_f32_sgpr); RUN(p_fma_f32_sgpr); RUN(p_add_f32_const); RUN(p_add_f32_lit); RUN(p_mov_b32_sgpr); RUN(p_sub_f32); RUN(p_add_u32); RUN(p_lshlrev); RUN(p_and_b32); RUN(p_fma_f32_neg); RUN(p_add_f32_e64abs); RUN(p_cvt_f32_i32); RUN(p_max3_f32);
    RUN(p_mov_b32); RUN(p_mov_b64); RUN(p_pk_mov_b32); RUN(p_lshl_add); RUN(p_s_nop);
    run("p_ds_read_b128", p_ds_read_b128, 8192); run("p_ds_read_b96", p_ds_read_b96, 8192);
    run("p_ds_read2_b32", p_ds_read2_b32, 8192); run("p_ds_read_b64", p_ds_read_b64, 8192);
    run("p_ds_write_b128", p_ds_write_b128, 8192); run("p_ds_write_b96", p_ds_write_b96, 8192);
    return 0;
}
