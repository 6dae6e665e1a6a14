/* hrnet_io.h - C ABI of libhrnet_io.so: the host-side input pipeline of HighRes-net.
 *
 * The reference loads every batch with skimage.io.imread + numpy + torch ops in DataLoader worker processes
 * (src/DataLoader.py:72-148 read_imageset, :170-204 ImagesetDataset.__getitem__, src/utils.py:63-113 collateFunction):
 * at B = 32, 32 views that is 1024 + 64 PNG decodes per step.  This library is the native replacement of the byte work:
 *
 *   hrn_io_png_info / hrn_io_png_read_u16   <-  skimage.io.imread(...) for the PROBA-V assets (8 / 16-bit grayscale PNG)
 *   hrn_io_collate                          <-  np.array([imread(LRi) ...], uint16), imread(SM) -> bool, imread(HR),
 *                                               get_patch (DataLoader.py:16-31, :130-139), skimage.img_as_float(...).astype(float32)
 *                                               (:195-199) and collateFunction's truncate / zero-pad to min_L with the 1/0
 *                                               alpha indicators (utils.py:85-95), written straight into caller-owned
 *                                               (typically pinned) batch buffers by a pool of threads
 *
 * What stays in Python (highres-net_amd/DataLoader.py, utils.py): directory listing, clearance.npy, the clearance-softmax
 * view sampling and the random patch position - they draw from numpy's global RNG exactly like the reference, and their
 * results are passed in here as explicit file lists and (x, y) corners.
 *
 * Host pointers only; no GPU, no torch.  Return 0 on success, negative on error (-2 bad argument, -4 I/O or format error);
 * hrn_io_last_error() gives a thread-local message.
 */
#ifndef HRNET_IO_H
#define HRNET_IO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int hrn_io_version(void);
const char* hrn_io_last_error(void);

/* Header of a PNG file: width, height, bit depth (1, 2, 4, 8, 16).  Grayscale, non-interlaced only. */
int hrn_io_png_info(const char* path, int* width, int* height, int* bit_depth);
/* Decode into out[height][width] (values as stored: 0..2^depth-1).  width / height must match the file. */
int hrn_io_png_read_u16(const char* path, uint16_t* out, int width, int height);

/* Collate n_sets imagesets into batch buffers.
 *   lr_paths    : all LR view files, imageset after imageset (sum of n_views entries), already in the order the views are used
 *   n_views     : views available per imageset; the first min(n_views, min_L) are used, the rest of the min_L slots are zero
 *   hr_paths    : HR.png per imageset, or NULL (whole argument) / NULL entries when there is no HR (test split)
 *   sm_paths    : SM.png per imageset (status map; any non-zero sample -> 1.0)
 *   lr_size     : side of the stored LR images (HR / SM are 3 * lr_size)
 *   patch       : 0 = whole images; > 0: LR[x:x+patch, y:y+patch] with x the ROW and y the COLUMN corner as in get_patch,
 *                 HR / SM[3x:3x+3patch, 3y:3y+3patch]; px / py hold one corner per imageset
 *   lrs         : out (n_sets, min_L, S, S) f32, S = patch ? patch : lr_size; value = (float)(u16 / 65535.0)
 *   alphas      : out (n_sets, min_L) f32
 *   hrs, maps   : out (n_sets, 3S, 3S) f32 (hrs may be NULL when hr_paths is NULL)
 *   n_threads   : worker threads (<= 0: hardware concurrency) */
int hrn_io_collate(int n_sets, const char* const* lr_paths, const int* n_views, const char* const* hr_paths,
                   const char* const* sm_paths, int min_L, int lr_size, int patch, const int* px, const int* py,
                   float* lrs, float* alphas, float* hrs, float* maps, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
