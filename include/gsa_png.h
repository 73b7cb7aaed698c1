/*
 * gsa_png.h -- C ABI of the on-device PNG compressor for the mask files of the dataset writer.
 *
 * The reference stores every mask with cv2.imwrite("mask_%06d.png", mask[:, :, 0]) (reference main.py:102-103):
 * an 8-bit greyscale PNG holding the class index, compressed by zlib at cv2's defaults (level 1, run-length
 * strategy).  PNG is lossless, so the contract is the decoded pixels: any conforming zlib stream will do.  Here the
 * mask that gsa_generate left in HBM is filtered (PNG filter type 2 "Up": a row minus the row above, so that rows
 * equal to their predecessor become zeros) and run-length coded on the GPU into deflate blocks with the FIXED Huffman
 * code (RFC 1951 section 3.2.6: literals + matches of distance 1), one independent lane per group of rows; every
 * group ends with an empty stored block, which byte-aligns it (what zlib's Z_SYNC_FLUSH emits), so groups concatenate
 * by bytes.  Only the compressed stream (~1 % of the pixels for blob masks) crosses PCIe.
 *
 * Conventions as include/gsa.h: `stream` is a hipStream_t as void*, calls are stream-ordered and never synchronise,
 * device pointers unless stated, 0 on success / negative gsa_status on error.  Stateless: no context.
 */
#ifndef GSA_PNG_H
#define GSA_PNG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bytes of device workspace gsa_png_encode needs for n masks. */
int64_t gsa_png_workspace_bytes(int32_t n, int32_t H, int32_t W);

/* Upper bound of one mask's zlib stream: an `out_stride` that can never overflow. */
int64_t gsa_png_max_stream_bytes(int32_t H, int32_t W);

/* Compress n masks (n,H,W) u8, W a multiple of 16, 16-byte aligned.  Mask i's complete zlib stream (2-byte header,
 * deflate blocks, Adler-32 of the filtered scanlines) -- the payload of the PNG's single IDAT chunk -- goes to
 * out + i*out_stride and lengths[i] = its byte count.  The file is: PNG signature, IHDR (W, H, bit depth 8, colour
 * type 0), IDAT(stream), IEND; the host adds the chunk framing and CRCs (gan-segmentation_amd/png.py).
 * If a stream needs more than out_stride bytes, lengths[i] = -(bytes needed) and its output is truncated. */
int gsa_png_encode(void* stream, int32_t n, int32_t H, int32_t W, const uint8_t* mask, void* workspace,
                   int64_t workspace_bytes, uint8_t* out, int64_t out_stride, int32_t* lengths);

#ifdef __cplusplus
}
#endif
#endif /* GSA_PNG_H */
