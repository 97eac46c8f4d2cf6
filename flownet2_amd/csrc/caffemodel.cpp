// .caffemodel reader: walks a serialized NetParameter (src/caffe/proto/caffe.proto:66-98) and yields, for every layer that
// carries trained blobs, (layer name, blob index, shape, float data) -- what Net::CopyTrainedLayersFrom (src/caffe/net.cpp:752-800)
// reads through libprotobuf + Blob::FromProto (src/caffe/blob.cpp:459-508).  Host code only, no allocation: the caller maps the
// file, asks for the index, then copies the blobs it wants.
//
// Wire format handled (proto2): NetParameter.layer = 100 (LayerParameter: name 1, type 2, blobs 7) and the deprecated
// NetParameter.layers = 2 (V1LayerParameter: name 4, type 5 = enum, blobs 6; the reference upgrades those files on load,
// util/upgrade_proto.cpp -- same blobs);  BlobProto: shape 7 (BlobShape.dim 1, packed or not), legacy num / channels / height /
// width 1-4, data 5 and double_data 8 (packed or one element per key; several chunks concatenate, as protobuf merges them).
#include <cstdint>
#include <cstring>

#include "fn2_common.hpp"

namespace {

struct Rd {
  const unsigned char* p; const unsigned char* end;
  bool varint(uint64_t* v) {
    uint64_t r = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      if (p >= end) return false;
      const unsigned char b = *p++;
      r |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) { *v = r; return true; }
    }
    return false;
  }
  // next field: key, and for length-delimited fields the payload [q, q + n)
  bool field(unsigned* num, unsigned* wt, uint64_t* val, const unsigned char** q, size_t* n) {
    uint64_t key;
    if (!varint(&key) || (key >> 3) == 0 || (key >> 3) > 0x1fffffffull) return false;
    *num = (unsigned)(key >> 3); *wt = (unsigned)(key & 7);
    switch (*wt) {
      case 0: return varint(val);
      case 1: if (end - p < 8) return false; *q = p; *n = 8; p += 8; return true;
      case 5: if (end - p < 4) return false; *q = p; *n = 4; p += 4; return true;
      case 2: {
        uint64_t l;
        if (!varint(&l) || (uint64_t)(end - p) < l) return false;
        *q = p; *n = (size_t)l; p += l; return true;
      }
      default: return false;     // groups do not occur in caffe.proto
    }
  }
};

// shape and element count of one BlobProto
bool blob_meta(const unsigned char* b, size_t len, fn2_caffemodel_entry* e) {
  Rd r{b, b + len};
  long long legacy[4] = {0, 0, 0, 0};
  bool has_shape = false, has_legacy = false;
  size_t nf = 0, nd = 0;
  e->num_axes = 0;
  while (r.p < r.end) {
    unsigned num, wt; uint64_t v = 0; const unsigned char* q = nullptr; size_t n = 0;
    if (!r.field(&num, &wt, &v, &q, &n)) return false;
    if (num >= 1 && num <= 4 && wt == 0) { legacy[num - 1] = (long long)(int32_t)v; has_legacy = true; }
    else if (num == 7 && wt == 2) {                       // BlobShape
      has_shape = true;
      Rd s{q, q + n};
      while (s.p < s.end) {
        unsigned sn, sw; uint64_t sv = 0; const unsigned char* sq = nullptr; size_t sl = 0;
        if (!s.field(&sn, &sw, &sv, &sq, &sl)) return false;
        if (sn != 1) continue;
        if (sw == 0) { if (e->num_axes >= 8) return false; e->dim[e->num_axes++] = (long long)sv; }
        else if (sw == 2) {
          Rd d{sq, sq + sl};
          while (d.p < d.end) { uint64_t x; if (!d.varint(&x) || e->num_axes >= 8) return false; e->dim[e->num_axes++] = (long long)x; }
        }
      }
    } else if (num == 5) { if (wt == 2) nf += n / 4; else if (wt == 5) nf += 1; }
    else if (num == 8) { if (wt == 2) nd += n / 8; else if (wt == 1) nd += 1; }
  }
  if (!has_shape && has_legacy) { e->num_axes = 4; for (int i = 0; i < 4; ++i) e->dim[i] = legacy[i]; }   // blob.cpp:462-471
  e->is_double = nd > 0;                                   // blob.cpp:481: double_data wins when present
  e->count = nd > 0 ? nd : nf;
  return true;
}

}  // namespace

using namespace fn2;

FN2_API int fn2_caffemodel_index(const void* buf, size_t len, fn2_caffemodel_entry* entries, int max_entries, int* num_entries) {
  if (!buf || !num_entries || (max_entries > 0 && !entries)) return fail(FN2_ERR_INVALID_ARG, "caffemodel_index: null argument");
  const unsigned char* base = static_cast<const unsigned char*>(buf);
  Rd r{base, base + len};
  int count = 0;
  while (r.p < r.end) {
    unsigned num, wt; uint64_t v = 0; const unsigned char* q = nullptr; size_t n = 0;
    if (!r.field(&num, &wt, &v, &q, &n)) return fail(FN2_ERR_INVALID_ARG, "caffemodel: malformed NetParameter at byte %zu", (size_t)(r.p - base));
    const bool v1 = num == 2 && wt == 2;
    if (!(num == 100 && wt == 2) && !v1) continue;
    const unsigned f_name = v1 ? 4 : 1, f_type = v1 ? 5 : 2, f_blobs = v1 ? 6 : 7;
    Rd l{q, q + n};
    size_t name_off = 0, name_len = 0, type_off = 0, type_len = 0;
    long long v1_type = -1;
    int blob_index = 0;
    const int first = count;
    while (l.p < l.end) {
      unsigned ln, lw; uint64_t lv = 0; const unsigned char* lq = nullptr; size_t ll = 0;
      if (!l.field(&ln, &lw, &lv, &lq, &ll)) return fail(FN2_ERR_INVALID_ARG, "caffemodel: malformed layer message at byte %zu", (size_t)(l.p - base));
      if (ln == f_name && lw == 2) { name_off = (size_t)(lq - base); name_len = ll; }
      else if (ln == f_type && lw == 2 && !v1) { type_off = (size_t)(lq - base); type_len = ll; }
      else if (ln == f_type && lw == 0 && v1) v1_type = (long long)lv;
      else if (ln == f_blobs && lw == 2) {
        if (count < max_entries) {
          fn2_caffemodel_entry* e = &entries[count];
          std::memset(e, 0, sizeof(*e));
          e->blob_index = blob_index; e->blob_off = (size_t)(lq - base); e->blob_len = ll; e->v1 = v1 ? 1 : 0;
          if (!blob_meta(lq, ll, e)) return fail(FN2_ERR_INVALID_ARG, "caffemodel: malformed BlobProto at byte %zu", (size_t)(lq - base));
        } else {
          fn2_caffemodel_entry tmp;
          if (!blob_meta(lq, ll, &tmp)) return fail(FN2_ERR_INVALID_ARG, "caffemodel: malformed BlobProto at byte %zu", (size_t)(lq - base));
        }
        ++count; ++blob_index;
      }
    }
    for (int i = first; i < count && i < max_entries; ++i) {     // the name may follow the blobs on the wire
      entries[i].name_off = name_off; entries[i].name_len = name_len;
      entries[i].type_off = type_off; entries[i].type_len = type_len; entries[i].v1_type = v1_type;
    }
  }
  *num_entries = count;
  return FN2_OK;
}

FN2_API int fn2_caffemodel_read_blob(const void* buf, size_t len, const fn2_caffemodel_entry* e, float* dst, size_t dst_floats) {
  if (!buf || !e || !dst) return fail(FN2_ERR_INVALID_ARG, "caffemodel_read_blob: null argument");
  if (e->blob_off > len || e->blob_len > len - e->blob_off) return fail(FN2_ERR_INVALID_ARG, "caffemodel_read_blob: entry outside the buffer");
  if (dst_floats < e->count) return fail(FN2_ERR_INVALID_ARG, "caffemodel_read_blob: destination holds %zu floats, the blob %zu", dst_floats, e->count);
  long long want = 1;
  for (int i = 0; i < e->num_axes; ++i) want *= e->dim[i];
  if ((size_t)want != e->count)                            // blob.cpp:486 CHECK_EQ(count_, proto.data_size())
    return fail(FN2_ERR_INVALID_ARG, "caffemodel: blob shape holds %lld elements, data %zu", want, e->count);
  const unsigned char* b = static_cast<const unsigned char*>(buf) + e->blob_off;
  Rd r{b, b + e->blob_len};
  size_t k = 0;
  const unsigned want_field = e->is_double ? 8u : 5u;
  while (r.p < r.end) {
    unsigned num, wt; uint64_t v = 0; const unsigned char* q = nullptr; size_t n = 0;
    if (!r.field(&num, &wt, &v, &q, &n)) return fail(FN2_ERR_INVALID_ARG, "caffemodel_read_blob: malformed BlobProto");
    if (num != want_field || (wt != 2 && wt != 5 && wt != 1)) continue;
    if (!e->is_double) {
      const size_t m = n / 4;
      if (k + m > e->count) return fail(FN2_ERR_INVALID_ARG, "caffemodel_read_blob: more data than counted");
      std::memcpy(dst + k, q, 4 * m);                     // little-endian IEEE floats on the wire
      k += m;
    } else {
      const size_t m = n / 8;
      if (k + m > e->count) return fail(FN2_ERR_INVALID_ARG, "caffemodel_read_blob: more data than counted");
      for (size_t i = 0; i < m; ++i) { double d; std::memcpy(&d, q + 8 * i, 8); dst[k + i] = (float)d; }   // blob.cpp:490
      k += m;
    }
  }
  return FN2_OK;
}
