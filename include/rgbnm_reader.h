/* rgbnm_reader.h -- C ABI of librgbnm_reader.so: host-side partial JPEG decode (entropy decode only).
 * Replaces the reference's pybind11 extension entry dct_manip.read_coefficients (dct_manip/dct_manip.cpp:152-178,
 * :98-150, :78-96; bound at :578-669 and called from datasets.py:287).  Plain pointers, caller-owned buffers,
 * 0 on success, negative on error with a message in `err` (libjpeg's formatted message for -2).
 */
#ifndef RGBNM_READER_H
#define RGBNM_READER_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RGBNM_RD_EOPEN (-1)  /* "Unable to open file for reading: <path>" (dct_manip.cpp:155-159) */
#define RGBNM_RD_EJPEG (-2)  /* libjpeg error_exit (dct_manip.cpp:24-41) */
#define RGBNM_RD_EARG (-3)
#define RGBNM_RD_ESHAPE (-4)

int rgbnm_reader_abi_version(void);
/* info17[0] = number of components; per component c: info17[1+4c..] = height_in_blocks, width_in_blocks,
 * downsampled_height, downsampled_width. */
int rgbnm_jpeg_info(const char* path, int32_t* info17, char* err, int errlen);
int rgbnm_jpeg_info_mem(const unsigned char* buf, size_t len, int32_t* info17, char* err, int errlen);
/* dim int32 [C][2]; quant int16 [C][64] natural order; Y int16 [Hb*Wb*64]; CbCr int16 [2*Hbc*Wbc*64] or NULL. */
int rgbnm_read_coefficients(const char* path, int32_t* dim, int16_t* quant, int16_t* Y, int16_t* CbCr, char* err,
                            int errlen);
int rgbnm_read_coefficients_mem(const unsigned char* buf, size_t len, int32_t* dim, int16_t* quant, int16_t* Y,
                                int16_t* CbCr, char* err, int errlen);
/* n same-shaped files decoded by `threads` pthreads straight into (pinned) batch buffers; grayscale files get zero
 * chroma and unit chroma tables.  Returns the number of failed files; status[i] holds each file's code. */
int rgbnm_read_coefficients_batch(const char* const* paths, int n, int threads, int Hb, int Wb, int Hbc, int Wbc,
                                  int16_t* Y, int16_t* CbCr, int16_t* quant, int32_t* status);
/* The same with a crop box per file (luma blocks (top, left, height, width), all even; chroma box = halved): only the box
 * is copied out of libjpeg's coefficient arrays, packed [height][width][64] at Ypacked + yoff[i] and
 * [2][height/2][width/2][64] at Cpacked + coff[i] (element offsets).  For loader.DCTBatchLoader(crop_on_host=True): the crop
 * of RandomResizedCrop_DCT (utils/custom_transforms.py:631-663) is taken on the host, before the H2D copy
 * (datasets.py:274-297, pipeline_utils.py:70-73 ship whole coefficient tensors). */
int rgbnm_read_coefficients_batch_crop(const char* const* paths, int n, int threads, int Hb, int Wb, int Hbc, int Wbc,
                                       const int32_t* box, const int64_t* yoff, const int64_t* coff, int16_t* Ypacked,
                                       int16_t* Cpacked, int16_t* quant, int32_t* status);
#ifdef __cplusplus
}
#endif
#endif
