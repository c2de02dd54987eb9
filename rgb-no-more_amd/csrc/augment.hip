// DCT-domain data path on device, batched: quant-table de-normalise + clamp, block crop,
// resize {x2, identity, /2} by DCT conversion matrices, horizontal flip, two RandAugment ops, ToRange.
// Reference (per sample, CPU, inside DataLoader workers): datasets.py:286-293, utils/custom_transforms.py
// (RandomResizedCrop_DCT :631-663, RandomFlip_DCT :926-942, RandAugment_dct/_apply_op_dct :944-1127, ToRange
// :436-454), utils/dct_ops.py (resize :436-580, flip :601-621, rotate90 :99-130, translate :748-774, cutout
// :776-815, DC photometric ops :817-914, solarize_add :653-679, midfreqaug :710-746, sharpblur :681-708).
//
// Integer / index work is bit exact; resize is fp32 (one LSB on exact .5 ties).  All random draws
// are made on the host and arrive as explicit per-sample parameters (rgbnm_aug_params).
//
//   kernel 1  dct_resize      one wave per 16x16 super-block: raw int16 coefficients -> dequant/clamp -> A.X.A^T
//                             (or A^T.P.A) -> round -> flipped int16 image [B][75264]           (HBM: ~0.6 MB/img)
//   kernel 2  dct_randaug     one 1024-thread workgroup per image, the whole 28x28(+2x14x14) block image lives in
//                             LDS (147 KB): ops are applied in order (geometric ones as register-staged gathers,
//                             photometric ones with an exact integer DC reduction), then ToRange -> fp32/bf16.
#include "common.h"
#include "internal.h"
#include "../../include/rgbnm.h"

#define AUG_NS aug28
#define AUG_S 28
#define AUG_LDS 1
#include "augment_body.inc"
#undef AUG_NS
#undef AUG_S
#undef AUG_LDS
#define AUG_NS aug32
#define AUG_S 32
#define AUG_LDS 0
#include "augment_body.inc"
#undef AUG_NS
#undef AUG_S
#undef AUG_LDS

extern "C" {

size_t rgbnm_dct_augment_workspace_ex(int B, int size) {
  return size == 32 ? aug32::workspace_bytes(B) : aug28::workspace_bytes(B);
}
size_t rgbnm_dct_augment_workspace(int B) { return aug28::workspace_bytes(B); }

int rgbnm_dct_augment_ex(const int16_t* Yq, const int16_t* CbCrq, const int16_t* quant, const rgbnm_aug_params* params_dev,
                         const rgbnm_aug_params* params_host, const float* conv16, const float* filters, void* outY,
                         void* outC, int out_dtype, int size, int B, int Hy, int Wy, int Hc, int Wc, int entry_clamp,
                         int nops, void* workspace, size_t workspace_bytes, void* stream) {
  if (!Yq || !quant || !params_dev || !params_host || !conv16 || !outY || !outC || !workspace || B <= 0) return RGBNM_EINVAL;
  if (nops < 0 || nops > 2 || (size != 28 && size != 32)) return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (size == 32)
    return aug32::launch(Yq, CbCrq, quant, params_dev, params_host, conv16, filters, outY, outC, out_dtype, B, Hy, Wy, Hc,
                         Wc, entry_clamp, nops, workspace, workspace_bytes, st);
  return aug28::launch(Yq, CbCrq, quant, params_dev, params_host, conv16, filters, outY, outC, out_dtype, B, Hy, Wy, Hc, Wc,
                       entry_clamp, nops, workspace, workspace_bytes, st);
}

int rgbnm_dct_augment_packed(const int16_t* Ypacked, const int16_t* Cpacked, const long long* y_off, const long long* c_off,
                             const int16_t* quant, const rgbnm_aug_params* params_dev, const rgbnm_aug_params* params_host,
                             const float* conv16, const float* filters, void* outY, void* outC, int out_dtype, int size, int B,
                             int Hy, int Wy, int Hc, int Wc, int entry_clamp, int nops, void* workspace, size_t workspace_bytes,
                             void* stream) {
  if (!Ypacked || !y_off || !c_off || !quant || !params_dev || !params_host || !conv16 || !outY || !outC || !workspace || B <= 0)
    return RGBNM_EINVAL;
  if (nops < 0 || nops > 2 || (size != 28 && size != 32)) return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (size == 32)
    return aug32::launch(Ypacked, Cpacked, quant, params_dev, params_host, conv16, filters, outY, outC, out_dtype, B, Hy, Wy, Hc,
                         Wc, entry_clamp, nops, workspace, workspace_bytes, st, y_off, c_off);
  return aug28::launch(Ypacked, Cpacked, quant, params_dev, params_host, conv16, filters, outY, outC, out_dtype, B, Hy, Wy, Hc, Wc,
                       entry_clamp, nops, workspace, workspace_bytes, st, y_off, c_off);
}

int rgbnm_dct_augment(const int16_t* Yq, const int16_t* CbCrq, const int16_t* quant, const rgbnm_aug_params* params_dev,
                      const rgbnm_aug_params* params_host, const float* conv16, const float* filters, void* outY,
                      void* outC, int out_dtype, int B, int Hy, int Wy, int Hc, int Wc, int entry_clamp, int nops,
                      void* workspace, size_t workspace_bytes, void* stream) {
  return rgbnm_dct_augment_ex(Yq, CbCrq, quant, params_dev, params_host, conv16, filters, outY, outC, out_dtype, 28, B, Hy,
                              Wy, Hc, Wc, entry_clamp, nops, workspace, workspace_bytes, stream);
}

#ifdef AUG_PROF
int rgbnm_debug_aug_prof(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(aug28::g_aug_prof), sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : -1;
}
int rgbnm_debug_aug_prof2(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(aug28::g_aug_prof2), sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
