/* Host-side partial JPEG reader (stays C on the host, per north_star): entropy-decode only and hand back the
 * quantised 8x8 DCT coefficient blocks + quantisation tables, with the return contract of the reference's
 * dct_manip.read_coefficients (dct_manip/dct_manip.cpp:78-178):
 *   dim   int32 [C][2]  = (downsampled_height, downsampled_width) per component        (:103-107)
 *   quant int16 [C][8][8] natural (not zig-zag) order, as libjpeg's quant_table->quantval (:94-95)
 *   Y     int16 [1][Hb][Wb][8][8], CbCr int16 [2][Hb_c][Wb_c][8][8] (absent for 1-component files), blocks copied
 *         row-major per component from jpeg_read_coefficients' virtual arrays                 (:84-92, :110-141)
 * Built as librgbnm_reader.so (gcc, links IJG libjpeg 9); plain C ABI, no torch, caller owns every buffer.
 * A pthread batch entry decodes many files in parallel straight into one batch tensor.
 */
#include <pthread.h>
#include <setjmp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <jpeglib.h>

#define RD_OK 0
#define RD_EOPEN (-1)    /* "Unable to open file for reading: <path>" (dct_manip.cpp:155-159) */
#define RD_EJPEG (-2)    /* libjpeg error_exit; message in err (dct_manip.cpp:24-41) */
#define RD_EARG (-3)
#define RD_ESHAPE (-4)   /* file does not have the expected component/grid shape (batch entry) */

struct rd_err {
  struct jpeg_error_mgr pub;
  jmp_buf jb;
  char msg[JMSG_LENGTH_MAX];
};

static void rd_error_exit(j_common_ptr cinfo) {
  struct rd_err* e = (struct rd_err*)cinfo->err;
  (*cinfo->err->format_message)(cinfo, e->msg);
  longjmp(e->jb, 1);
}

static void set_err(char* err, int errlen, const char* msg) {
  if (err && errlen > 0) {
    strncpy(err, msg, (size_t)errlen - 1);
    err[errlen - 1] = 0;
  }
}

/* info[0] = ncomp; then per component: height_in_blocks, width_in_blocks, downsampled_height, downsampled_width */
/* box (may be NULL) = (top, left, height, width) in luma blocks, all even: only that box is copied, packed
 * [height][width][64] into Y and [2][height/2][width/2][64] into CbCr (the chroma box is the luma box halved). */
static int read_impl(FILE* fp, const unsigned char* mem, size_t memlen, int32_t* info, int32_t* dim, int16_t* quant,
                     int16_t* Y, int16_t* CbCr, const int32_t* expect, const int32_t* box, char* err, int errlen) {
  struct jpeg_decompress_struct cinfo;
  struct rd_err jerr;
  memset(&cinfo, 0, sizeof(cinfo));
  cinfo.err = jpeg_std_error(&jerr.pub);
  jerr.pub.error_exit = rd_error_exit;
  jerr.msg[0] = 0;
  if (setjmp(jerr.jb)) {
    set_err(err, errlen, jerr.msg);
    jpeg_destroy_decompress(&cinfo);
    return RD_EJPEG;
  }
  jpeg_create_decompress(&cinfo);
  if (fp) jpeg_stdio_src(&cinfo, fp);
  else jpeg_mem_src(&cinfo, (unsigned char*)mem, (unsigned long)memlen);
  jpeg_read_header(&cinfo, TRUE);
  const int nc = cinfo.num_components;
  if (info) {
    info[0] = nc;
    for (int c = 0; c < nc && c < 4; ++c) {
      info[1 + 4 * c + 0] = (int32_t)cinfo.comp_info[c].height_in_blocks;
      info[1 + 4 * c + 1] = (int32_t)cinfo.comp_info[c].width_in_blocks;
      info[1 + 4 * c + 2] = (int32_t)cinfo.comp_info[c].downsampled_height;
      info[1 + 4 * c + 3] = (int32_t)cinfo.comp_info[c].downsampled_width;
    }
  }
  if (!Y) {  /* header only */
    jpeg_destroy_decompress(&cinfo);
    return RD_OK;
  }
  if (expect) {  /* batch entry: every file must fill the same slots */
    if ((nc != 1 && nc != 3) || (int)cinfo.comp_info[0].height_in_blocks != expect[0] ||
        (int)cinfo.comp_info[0].width_in_blocks != expect[1] ||
        (nc == 3 && ((int)cinfo.comp_info[1].height_in_blocks != expect[2] ||
                     (int)cinfo.comp_info[1].width_in_blocks != expect[3] ||
                     (int)cinfo.comp_info[2].height_in_blocks != expect[2] ||
                     (int)cinfo.comp_info[2].width_in_blocks != expect[3]))) {
      set_err(err, errlen, "coefficient grid differs from the batch shape");
      jpeg_destroy_decompress(&cinfo);
      return RD_ESHAPE;
    }
  }
  /* CbCr is ONE [2][Hbc][Wbc] tensor sized from component 1 (dct_manip.py; reference dct_manip.cpp:112-117 does the
   * same and overruns): component 2 must have component 1's grid, and more than three components have no slot. */
  if (CbCr && nc >= 3 && (cinfo.comp_info[2].height_in_blocks != cinfo.comp_info[1].height_in_blocks ||
                          cinfo.comp_info[2].width_in_blocks != cinfo.comp_info[1].width_in_blocks)) {
    set_err(err, errlen, "Cb and Cr have different block grids (unsupported sampling)");
    jpeg_destroy_decompress(&cinfo);
    return RD_ESHAPE;
  }
  if (dim)
    for (int c = 0; c < nc && c < 3; ++c) {
      dim[2 * c + 0] = (int32_t)cinfo.comp_info[c].downsampled_height;
      dim[2 * c + 1] = (int32_t)cinfo.comp_info[c].downsampled_width;
    }
  if (box) {
    const int sh = nc >= 3 ? 1 : 0;      /* a chroma plane, if any, must cover the halved box */
    if (box[0] < 0 || box[1] < 0 || box[2] <= 0 || box[3] <= 0 || ((box[0] | box[1] | box[2] | box[3]) & 1) ||
        box[0] + box[2] > (int)cinfo.comp_info[0].height_in_blocks || box[1] + box[3] > (int)cinfo.comp_info[0].width_in_blocks ||
        (sh && ((box[0] + box[2]) / 2 > (int)cinfo.comp_info[1].height_in_blocks ||
                (box[1] + box[3]) / 2 > (int)cinfo.comp_info[1].width_in_blocks))) {
      set_err(err, errlen, "crop box outside the coefficient grid (or not even)");
      jpeg_destroy_decompress(&cinfo);
      return RD_EARG;
    }
  }
  jvirt_barray_ptr* coefs = jpeg_read_coefficients(&cinfo);   /* entropy decode only: no IDCT, no colour conversion */
  for (int c = 0; c < nc && c < 3; ++c) {
    jpeg_component_info* ci = &cinfo.comp_info[c];
    /* rows r0 .. r0 + nr - 1, blocks c0 .. c0 + ncol - 1 of this component, packed [nr][ncol][64] */
    const JDIMENSION r0 = box ? (JDIMENSION)(c ? box[0] / 2 : box[0]) : 0, c0 = box ? (JDIMENSION)(c ? box[1] / 2 : box[1]) : 0;
    const JDIMENSION nr = box ? (JDIMENSION)(c ? box[2] / 2 : box[2]) : ci->height_in_blocks;
    const JDIMENSION ncol = box ? (JDIMENSION)(c ? box[3] / 2 : box[3]) : ci->width_in_blocks;
    int16_t* dst;
    if (c == 0) dst = Y;
    else {
      if (!CbCr) continue;
      dst = CbCr + (size_t)(c - 1) * nr * ncol * DCTSIZE2;
    }
    for (JDIMENSION r = 0; r < nr; ++r) {
      JBLOCKARRAY rows = (*cinfo.mem->access_virt_barray)((j_common_ptr)&cinfo, coefs[c], r0 + r, 1, FALSE);
      memcpy(dst + (size_t)r * ncol * DCTSIZE2, rows[0][c0], (size_t)ncol * DCTSIZE2 * sizeof(int16_t));
    }
    if (quant && ci->quant_table)
      for (int k = 0; k < DCTSIZE2; ++k) quant[c * DCTSIZE2 + k] = (int16_t)ci->quant_table->quantval[k];
  }
  jpeg_finish_decompress(&cinfo);
  jpeg_destroy_decompress(&cinfo);
  return RD_OK;
}

int rgbnm_reader_abi_version(void) { return 1; }

int rgbnm_jpeg_info(const char* path, int32_t* info17, char* err, int errlen) {
  if (!path || !info17) return RD_EARG;
  FILE* fp = fopen(path, "rb");
  if (!fp) {
    char m[1200];
    snprintf(m, sizeof(m), "Unable to open file for reading: %s", path);
    set_err(err, errlen, m);
    return RD_EOPEN;
  }
  const int rc = read_impl(fp, NULL, 0, info17, NULL, NULL, NULL, NULL, NULL, NULL, err, errlen);
  fclose(fp);
  return rc;
}

int rgbnm_jpeg_info_mem(const unsigned char* buf, size_t len, int32_t* info17, char* err, int errlen) {
  if (!buf || !info17) return RD_EARG;
  return read_impl(NULL, buf, len, info17, NULL, NULL, NULL, NULL, NULL, NULL, err, errlen);
}

/* dim [C][2], quant [C][64], Y [Hb*Wb*64], CbCr [2*Hbc*Wbc*64] (may be NULL for 1-component files) */
int rgbnm_read_coefficients(const char* path, int32_t* dim, int16_t* quant, int16_t* Y, int16_t* CbCr, char* err,
                            int errlen) {
  if (!path || !dim || !quant || !Y) return RD_EARG;
  FILE* fp = fopen(path, "rb");
  if (!fp) {
    char m[1200];
    snprintf(m, sizeof(m), "Unable to open file for reading: %s", path);
    set_err(err, errlen, m);
    return RD_EOPEN;
  }
  const int rc = read_impl(fp, NULL, 0, NULL, dim, quant, Y, CbCr, NULL, NULL, err, errlen);
  fclose(fp);
  return rc;
}

int rgbnm_read_coefficients_mem(const unsigned char* buf, size_t len, int32_t* dim, int16_t* quant, int16_t* Y,
                                int16_t* CbCr, char* err, int errlen) {
  if (!buf || !dim || !quant || !Y) return RD_EARG;
  return read_impl(NULL, buf, len, NULL, dim, quant, Y, CbCr, NULL, NULL, err, errlen);
}

/* ---- batch: n files of identical grid (e.g. 512x512 4:2:0 -> 64x64 / 32x32 blocks) decoded by `threads` pthreads
 * straight into batch tensors Y [n][Hb*Wb*64], CbCr [n][2*Hbc*Wbc*64], quant [n][3][64]; status[i] per file.
 * Grayscale files leave their CbCr slot zero-filled and replicate... no: quant rows 1,2 are set to 1. ---- */
struct batch_job {
  const char* const* paths;
  int n, next;
  int32_t expect[4];
  int16_t *Y, *CbCr, *quant;
  int32_t* status;
  const int32_t* box;            /* [n][4] or NULL: crop boxes (packed output at yoff / coff) */
  const int64_t *yoff, *coff;
  pthread_mutex_t mu;
};

static void* batch_worker(void* arg) {
  struct batch_job* j = (struct batch_job*)arg;
  const size_t ysz = (size_t)j->expect[0] * j->expect[1] * 64, csz = (size_t)2 * j->expect[2] * j->expect[3] * 64;
  for (;;) {
    pthread_mutex_lock(&j->mu);
    const int i = j->next++;
    pthread_mutex_unlock(&j->mu);
    if (i >= j->n) break;
    int32_t dim[6];
    int16_t* q = j->quant + (size_t)i * 192;
    for (int k = 0; k < 192; ++k) q[k] = 1;
    const int32_t* bx = j->box ? j->box + 4 * (size_t)i : NULL;
    int16_t* yd = bx ? j->Y + j->yoff[i] : j->Y + (size_t)i * ysz;
    int16_t* cd = bx ? j->CbCr + j->coff[i] : j->CbCr + (size_t)i * csz;
    if (bx) {
      /* the box is validated against the EXPECTED grid before anything is written through it (read_impl checks it again against
       * the file's own grid): a negative or oversized box from a C caller is RD_EARG, not an out-of-bounds memset */
      if (bx[0] < 0 || bx[1] < 0 || bx[2] <= 0 || bx[3] <= 0 || ((bx[0] | bx[1] | bx[2] | bx[3]) & 1) ||
          (long long)bx[0] + bx[2] > j->expect[0] || (long long)bx[1] + bx[3] > j->expect[1] || j->yoff[i] < 0 || j->coff[i] < 0) {
        j->status[i] = RD_EARG;
        continue;
      }
    }
    memset(cd, 0, (bx ? (size_t)2 * (bx[2] / 2) * (bx[3] / 2) * 64 : csz) * sizeof(int16_t));
    FILE* fp = fopen(j->paths[i], "rb");
    if (!fp) {
      j->status[i] = RD_EOPEN;
      continue;
    }
    j->status[i] = read_impl(fp, NULL, 0, NULL, dim, q, yd, cd, j->expect, bx, NULL, 0);
    fclose(fp);
  }
  return NULL;
}

static int batch_run(struct batch_job* jp, int threads);

int rgbnm_read_coefficients_batch(const char* const* paths, int n, int threads, int Hb, int Wb, int Hbc, int Wbc,
                                  int16_t* Y, int16_t* CbCr, int16_t* quant, int32_t* status) {
  if (!paths || n <= 0 || !Y || !CbCr || !quant || !status || Hb <= 0 || Wb <= 0) return RD_EARG;
  struct batch_job j;
  j.paths = paths; j.n = n; j.next = 0; j.box = NULL; j.yoff = j.coff = NULL;
  j.expect[0] = Hb; j.expect[1] = Wb; j.expect[2] = Hbc; j.expect[3] = Wbc;
  j.Y = Y; j.CbCr = CbCr; j.quant = quant; j.status = status;
  return batch_run(&j, threads);
}

/* The same with a crop box per file: only box[i] = (top, left, height, width) (luma blocks, even) of file i is copied, packed
 * [height][width][64] at Ypacked + yoff[i] and [2][height/2][width/2][64] at Cpacked + coff[i] (element offsets chosen by the
 * caller: typically the running sums of the box sizes, so that one H2D copy of the used prefix ships the whole batch). */
int rgbnm_read_coefficients_batch_crop(const char* const* paths, int n, int threads, int Hb, int Wb, int Hbc, int Wbc,
                                       const int32_t* box, const int64_t* yoff, const int64_t* coff, int16_t* Ypacked,
                                       int16_t* Cpacked, int16_t* quant, int32_t* status) {
  if (!paths || n <= 0 || !Ypacked || !Cpacked || !quant || !status || !box || !yoff || !coff || Hb <= 0 || Wb <= 0) return RD_EARG;
  struct batch_job j;
  j.paths = paths; j.n = n; j.next = 0; j.box = box; j.yoff = yoff; j.coff = coff;
  j.expect[0] = Hb; j.expect[1] = Wb; j.expect[2] = Hbc; j.expect[3] = Wbc;
  j.Y = Ypacked; j.CbCr = Cpacked; j.quant = quant; j.status = status;
  return batch_run(&j, threads);
}

static int batch_run(struct batch_job* jp, int threads) {
  struct batch_job j = *jp;
  const int n = j.n;
  int32_t* status = j.status;
  /* (expect / buffers set by the caller) */
  j.next = 0;
  pthread_mutex_init(&j.mu, NULL);
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  if (threads > n) threads = n;
  pthread_t th[256];
  int started[256];
  for (int t = 1; t < threads; ++t) started[t] = pthread_create(&th[t], NULL, batch_worker, &j) == 0;
  batch_worker(&j);               /* the caller's thread drains whatever the (possibly fewer) workers leave */
  for (int t = 1; t < threads; ++t)
    if (started[t]) pthread_join(th[t], NULL);
  pthread_mutex_destroy(&j.mu);
  int bad = 0;
  for (int i = 0; i < n; ++i) bad += status[i] != 0;
  return bad;
}
