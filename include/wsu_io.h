/* libwsu_io.so -- host-side input plumbing of the evaluate loop (plain C ABI, no GPU code, zlib only).
 *
 * Replaces, for a whole batch at a time, the per-image read of the reference's evaluate loop:
 *   src/_defs/imread.py:19-23  imread4_u8 = cv2.imread + cv2.cvtColor(BGR2GRAY), of which the UNet path uses plane 3 (Y)
 *   src/unet/evaluate.py:120   x = imread(fname)[..., 3:]
 * Supported: 8-bit gray and 8-bit RGB, non-interlaced PNG.  RGB uses cv2's fixed-point luma
 * (R*4899 + G*9617 + B*1868 + 8192) >> 14; gray is returned as stored (cv2 maps v,v,v -> v exactly).
 */
#ifndef WSU_IO_H
#define WSU_IO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define WSU_PNG_OK 0
#define WSU_PNG_IO (-1)           /* cannot open / read */
#define WSU_PNG_FORMAT (-2)       /* not a PNG or corrupt */
#define WSU_PNG_UNSUPPORTED (-3)  /* palette, alpha, 16 bit or interlaced: read it with another decoder */
#define WSU_PNG_SHAPE (-4)        /* height / width differ from the batch shape */

int wsu_io_version(void);

/* height / width of one file without decoding pixels; 0 or a WSU_PNG_* code */
int wsu_png_shape(const char* path, int* h, int* w);

/* Decode n files into dst[n][h][w] (uint8 Y planes; pass pinned memory to overlap the upload) on up to nthreads threads.
 * status[i] = 0 or the WSU_PNG_* code of file i (its plane is then untouched).  Returns the number of failed files, -1 on bad arguments. */
int wsu_png_read_luma_batch(const char* const* paths, int n, uint8_t* dst, int h, int w, int nthreads, int* status);

#ifdef __cplusplus
}
#endif
#endif
