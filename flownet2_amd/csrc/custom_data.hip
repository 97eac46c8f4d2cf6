// CustomData sample format: Datum wire format, the writer's packing and the decode of a batch of samples on the GPU.
//
// Reference: Datum (src/caffe/proto/caffe.proto:30-41), ImagePair::read_data (tools/convert_imageset_and_flow.cpp:142-206),
// DecodeData + CustomDataLayerPrefetch (src/caffe/layers/custom_data_layer.cpp:44-136, :209-300).  The reference decodes on one
// host thread into fp32 blobs and uploads those; here the packed bytes stay packed until they are in HBM:
//   per pixel 10.125 B are read (6 B images, 4 B flow, 1 bit occlusion) and 36 B written (9 fp32 planes) -- HBM-bound streaming,
//   one launch for the whole batch, 4 outputs per thread for the byte / int16 planes and 8 per thread for the bit plane.
#include "fn2_common.hpp"

#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace fn2 {

// ---------------------------------------------------------------------------------------------------------
// Protobuf wire format (proto2): varint (0), 64-bit (1), length-delimited (2), groups (3 / 4, only ever unknown fields here), 32-bit (5).
// ---------------------------------------------------------------------------------------------------------
struct Reader {
  const unsigned char* p;
  const unsigned char* end;
  bool varint(uint64_t* v) {
    uint64_t r = 0;
    for (int shift = 0; shift < 64 && p < end; shift += 7) {
      const unsigned char b = *p++;
      r |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) { *v = r; return true; }
    }
    return false;
  }
  bool skip(size_t n) { if ((size_t)(end - p) < n) return false; p += n; return true; }
};

// Skips the body of an unknown group (wire type 3) up to its END_GROUP key (wire type 4, same field number); groups nest.
static bool skip_group(Reader& r, uint32_t field, int depth) {
  if (depth > 100) return false;                       // libprotobuf's recursion limit
  while (r.p < r.end) {
    uint64_t key, x;
    if (!r.varint(&key) || key > 0xffffffffull || (key >> 3) == 0) return false;
    const uint32_t f = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
    switch (wt) {
      case 0: if (!r.varint(&x)) return false; break;
      case 1: if (!r.skip(8)) return false; break;
      case 2: if (!r.varint(&x) || (uint64_t)(r.end - r.p) < x) return false; r.p += x; break;
      case 3: if (!skip_group(r, f, depth + 1)) return false; break;
      case 4: return f == field;
      case 5: if (!r.skip(4)) return false; break;
      default: return false;
    }
  }
  return false;                                        // ran out of bytes inside the group
}

// Walks the message once; float_dst (may be NULL) receives up to float_cap values of field 6.
static int walk_datum(const void* buf, size_t len, fn2_datum_view* out, float* float_dst, size_t float_cap) {
  if (!buf && len) return fail(FN2_ERR_INVALID_ARG, "datum: NULL buffer");
  Reader r{static_cast<const unsigned char*>(buf), static_cast<const unsigned char*>(buf) + len};
  fn2_datum_view v;
  std::memset(&v, 0, sizeof(v));
  while (r.p < r.end) {
    uint64_t key;
    if (!r.varint(&key)) return fail(FN2_ERR_INVALID_ARG, "datum: truncated field key");
    if (key > 0xffffffffull) return fail(FN2_ERR_INVALID_ARG, "datum: field key does not fit 32 bits");
    const uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
    if (field == 0) return fail(FN2_ERR_INVALID_ARG, "datum: field number 0");
    uint64_t x = 0;
    switch (wt) {
      case 0:
        if (!r.varint(&x)) return fail(FN2_ERR_INVALID_ARG, "datum: truncated varint (field %u)", field);
        if (field == 1) v.channels = (int)(int64_t)x;
        else if (field == 2) v.height = (int)(int64_t)x;
        else if (field == 3) v.width = (int)(int64_t)x;
        else if (field == 5) v.label = (int)(int64_t)x;
        else if (field == 7) v.encoded = x != 0;
        break;
      case 1:
        if (!r.skip(8)) return fail(FN2_ERR_INVALID_ARG, "datum: truncated 64-bit field %u", field);
        break;
      case 2: {
        if (!r.varint(&x) || (uint64_t)(r.end - r.p) < x) return fail(FN2_ERR_INVALID_ARG, "datum: truncated length-delimited field %u", field);
        if (field == 4) { v.data = r.p; v.data_bytes = (size_t)x; }
        else if (field == 6) {                       // packed repeated float
          if (x % 4) return fail(FN2_ERR_INVALID_ARG, "datum: packed float_data of %llu bytes", (unsigned long long)x);
          for (uint64_t i = 0; i < x / 4; ++i, ++v.float_data_count)
            if (float_dst && v.float_data_count < float_cap) std::memcpy(float_dst + v.float_data_count, r.p + 4 * i, 4);
        }
        r.p += x;
        break;
      }
      case 3:                                        // an unknown group: skipped as a whole
        if (!skip_group(r, field, 1)) return fail(FN2_ERR_INVALID_ARG, "datum: malformed group (field %u)", field);
        break;
      case 5:
        if ((size_t)(r.end - r.p) < 4) return fail(FN2_ERR_INVALID_ARG, "datum: truncated 32-bit field %u", field);
        if (field == 6) {
          if (float_dst && v.float_data_count < float_cap) std::memcpy(float_dst + v.float_data_count, r.p, 4);
          ++v.float_data_count;
        }
        r.p += 4;
        break;
      default:                                       // 4 = END_GROUP without a group, 6 / 7 = undefined
        return fail(FN2_ERR_INVALID_ARG, "datum: unexpected wire type %u (field %u)", wt, field);
    }
  }
  if (out) *out = v;
  return FN2_OK;
}

static size_t varint_size(uint64_t v) { size_t n = 1; while (v >= 0x80) { v >>= 7; ++n; } return n; }
static unsigned char* put_varint(unsigned char* p, uint64_t v) {
  while (v >= 0x80) { *p++ = (unsigned char)(v | 0x80); v >>= 7; }
  *p++ = (unsigned char)v;
  return p;
}
// int32 fields are sign-extended to 64 bits on the wire
static uint64_t int32_wire(int v) { return (uint64_t)(int64_t)v; }

// ---------------------------------------------------------------------------------------------------------
// Slicing of a sample (DecodeData :66-86)
// ---------------------------------------------------------------------------------------------------------
struct Slice { int c0, cc, enc; size_t offset; };
constexpr int kMaxSlices = 32;

static int make_slices(int channels, int H, int W, const int* sp, int nsp, const int* enc, int nenc, int float_data,
                       Slice* out, size_t* total) {
  if (channels < 1 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "custom_data: bad datum shape [%d,%d,%d]", channels, H, W);
  if (nsp < 0 || nsp + 1 > kMaxSlices || nenc < 0 || (nsp && !sp) || (nenc && !enc)) return fail(FN2_ERR_INVALID_ARG, "custom_data: bad slice arrays");
  if (float_data && nenc) return fail(FN2_ERR_INVALID_ARG, "Encoded layers must be stored as uint8 in LMDB.");          // :55
  const size_t hw = (size_t)H * W;
  int prev = 0;
  size_t off = 0;
  for (int s = 0; s <= nsp; ++s) {
    const int end = (s == nsp) ? channels : sp[s];
    if (end <= prev || end > channels) return fail(FN2_ERR_INVALID_ARG, "custom_data: slice point %d not in (%d, %d]", end, prev, channels);   // CHECK_GT :519
    const int e = float_data ? 0 : (s < nenc ? enc[s] : FN2_ENC_UINT8);                                                  // :80-83
    out[s] = Slice{prev, end - prev, e, off};
    if (float_data) off += 4 * hw * (end - prev);
    else if (e == FN2_ENC_UINT8) off += hw * (end - prev);
    else if (e == FN2_ENC_UINT16FLOW) off += 2 * hw * (end - prev);
    else if (e == FN2_ENC_BOOL1) {
      if (end - prev != 1) return fail(FN2_ERR_INVALID_ARG, "custom_data: BOOL1 slice with %d channels (the reference decodes H*W bits per slice, assert :135)", end - prev);
      off += (hw - 1) / 8 + 1;
    } else return fail(FN2_ERR_INVALID_ARG, "Invalid format for slice %d", s);                                            // :130
    prev = end;
  }
  *total = off;
  return FN2_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Decode kernel: ONE launch for all slices of all samples (a batch of eight 512x384 samples is 72 MB of traffic; four separate
// launches of 6-9 us each spend most of their time ramping up and draining).  grid = (blocks of the largest slice, sample, slice);
// a block works on one slice only, so the format switch is uniform.  Each thread produces 4 consecutive elements (8 for bits).
// ---------------------------------------------------------------------------------------------------------
struct SliceArgs {
  size_t offset;               // byte offset of the slice inside a sample
  size_t count;                // cc * H * W elements of the slice per sample
  const float* mean;           // already offset to the slice's first channel, or NULL
  float* top;
  int enc;                     // FN2_ENC_*, 0 = fp32 payload
};
struct DecodeArgs {
  const unsigned char* samples;
  size_t stride;
  float scale;
  SliceArgs slice[kMaxSlices];
};

__device__ __forceinline__ float finish(float v, const float* mean, size_t i, float scale) {
  return (v - (mean ? mean[i] : 0.f)) * scale;                                                // :282
}

__device__ __forceinline__ float flow_value(unsigned lo, unsigned hi) {
  const short v = (short)(unsigned short)(lo | (hi << 8));                                     // :99-101 (little-endian host)
  // signaling NaN in the reference (:104-105), quieted by the subtraction of the mean: 0x7fe00000 on x86 SSE
  return v == 32767 ? __uint_as_float(0x7fa00000u) : (float)v / 32.0f;                         // :107
}

__device__ __forceinline__ void store4(float* dst, size_t i, size_t n, const float* o) {
  if (n == 4 && (reinterpret_cast<uintptr_t>(dst + i) & 15) == 0) *reinterpret_cast<float4*>(dst + i) = make_float4(o[0], o[1], o[2], o[3]);
  else for (size_t k = 0; k < n; ++k) dst[i + k] = o[k];
}

__global__ void __launch_bounds__(256) decode_samples(DecodeArgs a) {
  const SliceArgs& s = a.slice[blockIdx.z];
  const unsigned char* src = a.samples + (size_t)blockIdx.y * a.stride + s.offset;
  float* dst = s.top + (size_t)blockIdx.y * s.count;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s.enc == FN2_ENC_BOOL1) {                                                               // :113-128, LSB first
    const size_t i = t * 8;
    if (i >= s.count) return;
    const unsigned d = src[t];
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = ((d >> k) & 1u) ? 1.f : 0.f;
    for (int h = 0; h < 2; ++h) {
      const size_t j = i + 4 * h;
      if (j >= s.count) break;
      const size_t n = (j + 4 <= s.count) ? 4 : s.count - j;
      float q[4];
      for (size_t k = 0; k < n; ++k) q[k] = finish(o[4 * h + k], s.mean, j + k, a.scale);
      store4(dst, j, n, q);
    }
    return;
  }
  const size_t i = t * 4;
  if (i >= s.count) return;
  const size_t n = (i + 4 <= s.count) ? 4 : s.count - i;
  float o[4];
  if (s.enc == FN2_ENC_UINT8) {                                                               // :88-92
    if (n == 4 && (reinterpret_cast<uintptr_t>(src + i) & 3) == 0) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(src + i);
      o[0] = (float)(w & 0xff); o[1] = (float)((w >> 8) & 0xff); o[2] = (float)((w >> 16) & 0xff); o[3] = (float)(w >> 24);
    } else {
      for (size_t k = 0; k < n; ++k) o[k] = (float)src[i + k];
    }
  } else if (s.enc == FN2_ENC_UINT16FLOW) {                                                   // :94-111
    if (n == 4 && (reinterpret_cast<uintptr_t>(src + 2 * i) & 7) == 0) {
      const uint2 w = *reinterpret_cast<const uint2*>(src + 2 * i);
      o[0] = flow_value(w.x & 0xff, (w.x >> 8) & 0xff); o[1] = flow_value((w.x >> 16) & 0xff, w.x >> 24);
      o[2] = flow_value(w.y & 0xff, (w.y >> 8) & 0xff); o[3] = flow_value((w.y >> 16) & 0xff, w.y >> 24);
    } else {
      for (size_t k = 0; k < n; ++k) o[k] = flow_value(src[2 * (i + k)], src[2 * (i + k) + 1]);
    }
  } else {                                                                                    // fp32 payload (Datum.float_data, :57-58)
    const float* f = reinterpret_cast<const float*>(src);
    for (size_t k = 0; k < n; ++k) o[k] = f[i + k];
  }
  for (size_t k = 0; k < n; ++k) o[k] = finish(o[k], s.mean, i + k, a.scale);
  store4(dst, i, n, o);
}

}  // namespace fn2

using namespace fn2;

FN2_API int fn2_datum_parse(const void* buf, size_t len, fn2_datum_view* out) {
  if (!out) return fail(FN2_ERR_INVALID_ARG, "datum_parse: out == NULL");
  return walk_datum(buf, len, out, nullptr, 0);
}

FN2_API int fn2_datum_float_data(const void* buf, size_t len, float* dst, size_t count) {
  if (!dst && count) return fail(FN2_ERR_INVALID_ARG, "datum_float_data: dst == NULL");
  fn2_datum_view v;
  int rc = walk_datum(buf, len, &v, dst, count);
  if (rc) return rc;
  if (v.float_data_count != count) return fail(FN2_ERR_INVALID_ARG, "datum_float_data: datum holds %zu floats, %zu requested", v.float_data_count, count);
  return FN2_OK;
}

FN2_API long long fn2_datum_serialize(int channels, int height, int width, const void* data, size_t data_bytes, int label,
                                      void* dst, size_t dst_bytes) {
  if (!data && data_bytes) return fail(FN2_ERR_INVALID_ARG, "datum_serialize: NULL data");
  const size_t need = 1 + varint_size(int32_wire(channels)) + 1 + varint_size(int32_wire(height)) + 1 + varint_size(int32_wire(width)) +
                      1 + varint_size(data_bytes) + data_bytes + 1 + varint_size(int32_wire(label));
  if (!dst) return (long long)need;
  if (dst_bytes < need) return fail(FN2_ERR_WORKSPACE, "datum_serialize: %zu bytes needed, %zu given", need, dst_bytes);
  unsigned char* p = static_cast<unsigned char*>(dst);
  *p++ = 0x08; p = put_varint(p, int32_wire(channels));     // field 1, varint
  *p++ = 0x10; p = put_varint(p, int32_wire(height));       // field 2
  *p++ = 0x18; p = put_varint(p, int32_wire(width));        // field 3
  *p++ = 0x22; p = put_varint(p, data_bytes);               // field 4, length-delimited
  if (data_bytes) std::memcpy(p, data, data_bytes);
  p += data_bytes;
  *p++ = 0x28; p = put_varint(p, int32_wire(label));        // field 5
  return (long long)need;
}

FN2_API size_t fn2_custom_data_sample_bytes(int channels, int H, int W, const int* slice_points, int n_slice_points,
                                            const int* encodings, int n_encodings) {
  Slice sl[kMaxSlices];
  size_t total = 0;
  if (make_slices(channels, H, W, slice_points, n_slice_points, encodings, n_encodings, 0, sl, &total)) return 0;
  return total;
}

FN2_API int fn2_custom_data_encode_sample(const unsigned char* img0, const unsigned char* img1, const float* flow,
                                          const unsigned char* occ, int H, int W, unsigned char* dst, size_t dst_bytes) {
  if (H < 1 || W < 1 || !img0 || !img1 || !dst) return fail(FN2_ERR_INVALID_ARG, "custom_data_encode_sample: bad arguments");
  const size_t hw = (size_t)H * W, need = 10 * hw + (hw - 1) / 8 + 1;                          // :142-145
  if (dst_bytes < need) return fail(FN2_ERR_WORKSPACE, "custom_data_encode_sample: %zu bytes needed, %zu given", need, dst_bytes);
  std::memset(dst, 0, need);                                                                  // :147
  unsigned char* p = dst;
  for (const unsigned char* img : {img0, img1})                                               // :151-166
    for (int c = 0; c < 3; ++c)
      for (size_t i = 0; i < hw; ++i) *p++ = img[i * 3 + c];
  for (size_t j = 0; j < 2 * hw; ++j) {                                                       // :169-181
    short value = 0;
    if (flow) {
      if (flow[j] != flow[j]) value = std::numeric_limits<short>::max();
      else {
        // `short value = flo*32`: conversion toward zero; out-of-range products are undefined in C++ -- clamped here
        const float t = flow[j] * 32;
        value = t >= 32767.f ? (short)32767 : (t <= -32768.f ? (short)-32768 : (short)t);
      }
    }
    *p++ = (unsigned char)((unsigned short)value & 0xff);
    *p++ = (unsigned char)((unsigned short)value >> 8);
  }
  unsigned char current = 0;                                                                  // :185-203
  int idx = 0;
  for (size_t i = 0; i < hw; ++i) {
    if (occ && occ[i] > 0) current |= (unsigned char)(1u << idx);
    if (++idx == 8) { *p++ = current; idx = 0; current = 0; }
  }
  if (idx > 0) *p++ = current;
  return FN2_OK;
}

FN2_API int fn2_custom_data_stage_records(const void* const* records, const size_t* record_bytes, int N, void* staging, size_t sample_stride,
                                          int* channels, int* height, int* width, size_t* sample_bytes, int* labels) {
  if (N < 1 || !records || !record_bytes) return fail(FN2_ERR_INVALID_ARG, "custom_data_stage_records: no records");
  std::vector<fn2_datum_view> v((size_t)N);
  for (int i = 0; i < N; ++i) {
    int rc = walk_datum(records[i], record_bytes[i], &v[i], nullptr, 0);
    if (rc) return rc;
    if (!v[i].data) return fail(FN2_ERR_INVALID_ARG, "custom_data_stage_records: record %d holds no data bytes", i);
    if (v[i].channels != v[0].channels || v[i].height != v[0].height || v[i].width != v[0].width || v[i].data_bytes != v[0].data_bytes)
      return fail(FN2_ERR_INVALID_ARG, "custom_data_stage_records: record %d is [%d,%d,%d] with %zu bytes, record 0 is [%d,%d,%d] with %zu",
                  i, v[i].channels, v[i].height, v[i].width, v[i].data_bytes, v[0].channels, v[0].height, v[0].width, v[0].data_bytes);
  }
  if (channels) *channels = v[0].channels;
  if (height) *height = v[0].height;
  if (width) *width = v[0].width;
  if (sample_bytes) *sample_bytes = v[0].data_bytes;
  if (labels) for (int i = 0; i < N; ++i) labels[i] = v[i].label;
  if (!staging) return FN2_OK;
  if (sample_stride < v[0].data_bytes) return fail(FN2_ERR_WORKSPACE, "custom_data_stage_records: stride %zu < sample size %zu", sample_stride, v[0].data_bytes);
  auto copy = [&](int i) { std::memcpy(static_cast<unsigned char*>(staging) + (size_t)i * sample_stride, v[i].data, v[i].data_bytes); };
  if ((size_t)N * v[0].data_bytes < (1u << 20) || N == 1) {
    for (int i = 0; i < N; ++i) copy(i);
  } else {                                             // a few MB per record: one thread per record (up to 8) saturates the host memory better
    const int nt = N < 8 ? N : 8;
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { for (int i = t; i < N; i += nt) copy(i); });
    for (auto& x : th) x.join();
  }
  return FN2_OK;
}

FN2_API int fn2_custom_data_decode_forward(const void* samples, size_t sample_stride, int N, int channels, int H, int W,
                                           const int* slice_points, int n_slice_points, const int* encodings, int n_encodings,
                                           int float_data, const float* mean, float scale, float* const* tops, void* stream) {
  Slice sl[kMaxSlices];
  size_t total = 0;
  int rc = make_slices(channels, H, W, slice_points, n_slice_points, encodings, n_encodings, float_data, sl, &total);
  if (rc) return rc;
  if (N < 0) return fail(FN2_ERR_INVALID_ARG, "custom_data_decode: N < 0");
  if (N == 0) return FN2_OK;
  if (!samples || !tops) return fail(FN2_ERR_INVALID_ARG, "custom_data_decode: NULL pointer");
  if (sample_stride < total) return fail(FN2_ERR_INVALID_ARG, "custom_data_decode: sample stride %zu < sample size %zu", sample_stride, total);
  if (float_data && sample_stride % 4) return fail(FN2_ERR_INVALID_ARG, "custom_data_decode: float samples need a stride that is a multiple of 4");
  if (N > 65535) return fail(FN2_ERR_INVALID_ARG, "custom_data_decode: batch of %d samples (max 65535)", N);
  const size_t hw = (size_t)H * W;
  DecodeArgs a{};
  a.samples = static_cast<const unsigned char*>(samples);
  a.stride = sample_stride;
  a.scale = scale;
  size_t max_threads = 0;
  for (int s = 0; s <= n_slice_points; ++s) {
    if (!tops[s]) return fail(FN2_ERR_INVALID_ARG, "custom_data_decode: top[%d] == NULL", s);
    a.slice[s] = SliceArgs{sl[s].offset, (size_t)sl[s].cc * hw, mean ? mean + (size_t)sl[s].c0 * hw : nullptr, tops[s], sl[s].enc};
    const size_t per_thread = sl[s].enc == FN2_ENC_BOOL1 ? 8 : 4;
    const size_t threads = (a.slice[s].count + per_thread - 1) / per_thread;
    if (threads > max_threads) max_threads = threads;
  }
  const size_t blocks = (max_threads + 255) / 256;
  if (blocks > 0x7fffffffu) return fail(FN2_ERR_INVALID_ARG, "custom_data_decode: slice of %zu elements is too large", max_threads * 4);
  hipLaunchKernelGGL(decode_samples, dim3((unsigned)blocks, (unsigned)N, (unsigned)(n_slice_points + 1)), dim3(256), 0, as_stream(stream), a);
  return check_launch("custom_data_decode_forward");
}
