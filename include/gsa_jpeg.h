/*
 * gsa_jpeg.h -- C ABI of the on-device baseline JPEG encoder of the dataset writer.
 *
 * The reference stores every generated image with cv2.imwrite("img_%06d.jpg", img[:, :, ::-1]) on its single
 * Python thread (reference main.py:100-101): libjpeg(-turbo) at its defaults -- quality 95, YCbCr 4:2:0, integer
 * "islow" DCT, Annex K Huffman tables.  At the rate one MI355X now produces pairs this encode is the bottleneck of
 * `main.py generate` (14 ms of a host core per 1024^2 image), so it moves onto the GPU: the uint8 image that
 * gsa_generate left in HBM is encoded where it lies and only the compressed bytes (~1/7 of the pixels) cross PCIe.
 *
 * The arithmetic is libjpeg's, integer for integer (oracle/c/jpeg_oracle.c restates it and is pinned byte for byte
 * against libjpeg-turbo through Pillow): the file decodes to exactly what the reference's cv2 call would have stored
 * for the same pixels.  The one difference is structural: the scan carries a restart marker every `restart` MCUs
 * (DRI segment in the header), which is what makes the entropy coding parallel; any baseline decoder reads it.
 *
 * Conventions as include/gsa.h: `stream` is a hipStream_t as void*, calls are stream-ordered and never synchronise,
 * device pointers unless stated, 0 on success / negative gsa_status on error.  Stateless: no context.
 */
#ifndef GSA_JPEG_H
#define GSA_JPEG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* File header for an H x W image (HOST buffer): SOI, JFIF APP0, DQT x2, SOF0 (2x2,1x1,1x1 sampling), DHT x4,
 * DRI (when restart > 0), SOS -- everything in front of the entropy-coded data, the bytes libjpeg writes.
 * Returns the header length (629 with DRI); nothing is written past `cap`.  Negative on bad arguments. */
int64_t gsa_jpeg_header(int32_t H, int32_t W, int32_t quality, int32_t restart, uint8_t* host_buf, int64_t cap);

/* Bytes of device workspace gsa_jpeg_encode needs for n images (quantised coefficients + per-interval scratch). */
int64_t gsa_jpeg_workspace_bytes(int32_t n, int32_t H, int32_t W, int32_t restart);

/* Upper bound of one image's scan (entropy-coded data + restart markers + EOI): an `out_stride` that can never
 * overflow.  Typical q95 output is ~15 % of H*W*3; a smaller stride is allowed (see lengths). */
int64_t gsa_jpeg_max_scan_bytes(int32_t H, int32_t W, int32_t restart);

/* Encode n RGB images (n,H,W,3) u8, H and W multiples of 16 (every StyleGAN resolution >= 16 px is), 16-byte
 * aligned.  restart = MCUs (16x16 px) per restart interval, 1..65535 (up to 8: one wave per interval, the fast path;
 * longer intervals are coded by one lane each).  Image i's scan + EOI marker goes to
 * out + i*out_stride and lengths[i] = its byte count; the file is gsa_jpeg_header()'s bytes followed by those.
 * If an image needs more than out_stride bytes, lengths[i] = -(bytes needed) and its output is truncated. */
int gsa_jpeg_encode(void* stream, int32_t n, int32_t H, int32_t W, const uint8_t* rgb, int32_t quality, int32_t restart,
                    void* workspace, int64_t workspace_bytes, uint8_t* out, int64_t out_stride, int32_t* lengths);

#ifdef __cplusplus
}
#endif
#endif /* GSA_JPEG_H */
