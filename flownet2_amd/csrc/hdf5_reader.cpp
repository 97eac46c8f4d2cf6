// .caffemodel.h5 reader: a native walk of the HDF5 file format (no libhdf5 in the image), enough for what
// Net::CopyTrainedLayersFromHDF5 (src/caffe/net.cpp:823-882) and hdf5_load_nd_dataset (src/caffe/util/hdf5.cpp:9-79) read, and for
// what Net::ToHDF5 (net.cpp:896-950) + hdf5_save_nd_dataset (util/hdf5.cpp:81-123: H5LTmake_dataset_float = contiguous IEEE floats)
// write with libhdf5 1.8 / 1.10 defaults: groups `data/<layer name>` holding datasets `0`, `1`, ...
//
// Handled (HDF5 File Format Specification 2.0 / 3.0):
//   superblock v0 / v1 (root symbol-table entry) and v2 / v3 (root object header address), at offset 0 or behind a user block;
//   object headers v1 and v2 ("OHDR") with continuation blocks; old-style groups (Symbol Table message -> v1 B-tree "TREE" of "SNOD"
//   nodes + local "HEAP" names) and new-style groups with COMPACT link storage (Link messages); DENSE link storage (fractal heap) is
//   refused with a message -- libhdf5 only writes it for libver=latest files with more than 8 links per group;
//   dataspace v1 / v2 (simple, scalar), datatype classes fixed-point (1, 2, 4, 8 bytes, signed or not, either byte order) and IEEE
//   floating point (4 / 8 bytes, either byte order) -- H5LTread_dataset_float converts both to native float, so does this reader;
//   layout v1-v4: compact, contiguous, chunked (v1-v3: v1 B-tree of chunks; unallocated chunks read as zero); filter pipeline v1 / v2 with
//   deflate (1) and shuffle (2) -- the reference's own fixture src/caffe/test/test_data/sample_data_2_gzip.h5 is gzip + chunked.
// Host code.  The caller maps the file and passes the buffer; the index lists every dataset with its absolute path in name order (the
// order H5Lget_name_by_idx(H5_INDEX_NAME) gives the reference, util/hdf5.cpp:168-181).
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "fn2_common.hpp"

namespace {

constexpr uint64_t UNDEF = ~0ull;

struct File {
  const unsigned char* b; size_t len;
  uint64_t base = 0;
  int so = 8, sl = 8;           // size of offsets / lengths
  std::string err;
  bool fail(const std::string& m) { if (err.empty()) err = m; return false; }
  bool ok(uint64_t off, uint64_t n) const { return off <= len && n <= len - off; }
  uint64_t rd(uint64_t off, int n) const {       // little-endian unsigned of n bytes (caller checked the range)
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) v |= (uint64_t)b[off + i] << (8 * i);
    return v;
  }
  uint64_t addr(uint64_t off) const {            // a file address field: all ones = undefined
    uint64_t v = rd(off, so);
    const uint64_t ones = so == 8 ? ~0ull : ((1ull << (8 * so)) - 1);
    return v == ones ? UNDEF : v + base;
  }
};

struct Msg { unsigned type; uint64_t off; uint64_t size; };

// all messages of the object header at `at` (v1 or v2), continuation blocks followed
bool header_messages(File& f, uint64_t at, std::vector<Msg>* out) {
  if (!f.ok(at, 16)) return f.fail("HDF5: object header outside the file");
  if (std::memcmp(f.b + at, "OHDR", 4) == 0) {                 // version 2
    if (f.b[at + 4] != 2) return f.fail("HDF5: unknown object header version");
    const unsigned flags = f.b[at + 5];
    uint64_t p = at + 6;
    if (flags & 0x20) p += 16;
    if (flags & 0x10) p += 4;
    const int szb = 1 << (flags & 3);
    if (!f.ok(p, szb)) return f.fail("HDF5: truncated object header");
    const uint64_t chunk0 = f.rd(p, szb);
    p += szb;
    struct Blk { uint64_t off, size; };
    std::vector<Blk> blocks{{p, chunk0}};
    const int hdr = 4 + ((flags & 0x04) ? 2 : 0);
    for (size_t bi = 0; bi < blocks.size(); ++bi) {
      uint64_t q = blocks[bi].off, end = blocks[bi].off + blocks[bi].size;
      if (!f.ok(q, blocks[bi].size + 4)) return f.fail("HDF5: object header chunk outside the file");
      while (q + hdr <= end) {
        const unsigned type = f.b[q];
        const uint64_t size = f.rd(q + 1, 2);
        q += hdr;
        if (q + size > end) return f.fail("HDF5: object header message overruns its chunk");
        if (type == 0x10) {
          if (size < (uint64_t)(f.so + f.sl)) return f.fail("HDF5: short continuation message");
          const uint64_t coff = f.addr(q), clen = f.rd(q + f.so, f.sl);
          if (!f.ok(coff, clen) || clen < 8 || std::memcmp(f.b + coff, "OCHK", 4) != 0) return f.fail("HDF5: bad continuation block");
          blocks.push_back({coff + 4, clen - 8});
        } else if (type != 0) {
          out->push_back({type, q, size});
        }
        q += size;
      }
      if (blocks.size() > 4096) return f.fail("HDF5: continuation loop");
    }
    return true;
  }
  if (f.b[at] != 1) return f.fail("HDF5: unknown object header version");
  const unsigned nmsg = (unsigned)f.rd(at + 2, 2);
  const uint64_t hsize = f.rd(at + 8, 4);
  struct Blk { uint64_t off, size; };
  std::vector<Blk> blocks{{at + 16, hsize}};
  unsigned seen = 0;
  for (size_t bi = 0; bi < blocks.size() && seen < nmsg; ++bi) {
    uint64_t q = blocks[bi].off, end = blocks[bi].off + blocks[bi].size;
    if (!f.ok(q, blocks[bi].size)) return f.fail("HDF5: object header block outside the file");
    while (q + 8 <= end && seen < nmsg) {
      const unsigned type = (unsigned)f.rd(q, 2);
      const uint64_t size = f.rd(q + 2, 2);
      q += 8;
      if (q + size > end) return f.fail("HDF5: object header message overruns its block");
      ++seen;
      if (type == 0x10) {
        if (size < (uint64_t)(f.so + f.sl)) return f.fail("HDF5: short continuation message");
        blocks.push_back({f.addr(q), f.rd(q + f.so, f.sl)});
      } else if (type != 0) {
        out->push_back({type, q, size});
      }
      q += size;
    }
    if (blocks.size() > 4096) return f.fail("HDF5: continuation loop");
  }
  return true;
}

struct Link { std::string name; uint64_t header; };

// names of an old-style group: v1 B-tree (node type 0) over symbol-table nodes, names in the local heap
bool btree_group(File& f, uint64_t node, uint64_t heap_data, uint64_t heap_size, std::vector<Link>* out, int depth) {
  if (depth > 32 || !f.ok(node, 8 + 2 * f.so)) return f.fail("HDF5: group B-tree node outside the file");
  if (std::memcmp(f.b + node, "SNOD", 4) == 0) {
    const unsigned n = (unsigned)f.rd(node + 6, 2);
    const uint64_t esz = 2 * f.so + 4 + 4 + 16;
    if (!f.ok(node + 8, n * esz)) return f.fail("HDF5: symbol table node outside the file");
    for (unsigned i = 0; i < n; ++i) {
      const uint64_t e = node + 8 + i * esz;
      const uint64_t noff = f.rd(e, f.so);
      if (noff >= heap_size) return f.fail("HDF5: link name outside the local heap");
      const char* s = (const char*)f.b + heap_data + noff;
      const size_t maxn = (size_t)(heap_size - noff);
      const size_t l = strnlen(s, maxn);
      if (l == maxn) return f.fail("HDF5: unterminated link name");
      out->push_back({std::string(s, l), f.addr(e + f.so)});
    }
    return true;
  }
  if (std::memcmp(f.b + node, "TREE", 4) != 0 || f.b[node + 4] != 0) return f.fail("HDF5: bad group B-tree node");
  const unsigned used = (unsigned)f.rd(node + 6, 2);
  uint64_t p = node + 8 + 2 * f.so;
  if (!f.ok(p, (uint64_t)used * (f.sl + f.so) + f.sl)) return f.fail("HDF5: group B-tree node outside the file");
  for (unsigned i = 0; i < used; ++i) {
    p += f.sl;                                  // key i
    if (!btree_group(f, f.addr(p), heap_data, heap_size, out, depth + 1)) return false;
    p += f.so;
  }
  return true;
}

bool group_links(File& f, const std::vector<Msg>& msgs, std::vector<Link>* out, bool* is_group) {
  *is_group = false;
  for (const Msg& m : msgs) {
    if (m.type == 0x11) {                       // Symbol Table message
      *is_group = true;
      if (m.size < (uint64_t)(2 * f.so)) return f.fail("HDF5: short symbol table message");
      const uint64_t bt = f.addr(m.off), heap = f.addr(m.off + f.so);
      if (!f.ok(heap, 8 + 2 * f.sl + f.so) || std::memcmp(f.b + heap, "HEAP", 4) != 0) return f.fail("HDF5: bad local heap");
      const uint64_t hsize = f.rd(heap + 8, f.sl), hdata = f.addr(heap + 8 + 2 * f.sl);
      if (!f.ok(hdata, hsize)) return f.fail("HDF5: local heap data outside the file");
      if (!btree_group(f, bt, hdata, hsize, out, 0)) return false;
    } else if (m.type == 0x02) {                // Link Info: dense storage?
      *is_group = true;
      if (m.size < 2) return f.fail("HDF5: short link info message");
      const unsigned flags = f.b[m.off + 1];
      const uint64_t p = m.off + 2 + ((flags & 1) ? 8 : 0);
      if (!f.ok(p, f.so)) return f.fail("HDF5: short link info message");
      if (f.addr(p) != UNDEF)
        return f.fail("HDF5: group with dense link storage (fractal heap) is not supported; re-save with default libver bounds");
    } else if (m.type == 0x06) {                // Link message (compact storage)
      *is_group = true;
      uint64_t p = m.off;
      const uint64_t end = m.off + m.size;
      if (m.size < 4 || f.b[p] != 1) return f.fail("HDF5: unknown link message version");
      const unsigned flags = f.b[p + 1];
      p += 2;
      unsigned ltype = 0;
      if (flags & 0x08) ltype = f.b[p++];
      if (flags & 0x04) p += 8;
      if (flags & 0x10) p += 1;
      const int lsz = 1 << (flags & 3);
      if (p + lsz > end) return f.fail("HDF5: short link message");
      const uint64_t nlen = f.rd(p, lsz);
      p += lsz;
      if (p + nlen > end) return f.fail("HDF5: short link message");
      std::string name((const char*)f.b + p, (size_t)nlen);
      p += nlen;
      if (ltype != 0) continue;                 // soft / external links: H5LTfind_dataset would follow them; Caffe never writes them
      if (p + f.so > end) return f.fail("HDF5: short link message");
      out->push_back({name, f.addr(p)});
    }
  }
  std::sort(out->begin(), out->end(), [](const Link& a, const Link& b) { return a.name < b.name; });
  return true;
}

struct Dataset {
  int rank = -1; long long dim[8];
  int tclass = -1, tsize = 0; bool tsigned = false, big = false;
  int layout = -1;
  uint64_t addr = UNDEF, size = 0;            // contiguous / compact payload; chunked: B-tree address
  int cdims = 0; uint64_t chunk[9];           // chunk shape (elements) + element size
  struct Filter { unsigned id; std::vector<uint32_t> cd; };
  std::vector<Filter> filters;
  bool has_space = false, has_type = false, has_layout = false;
};

bool parse_dataset(File& f, const std::vector<Msg>& msgs, Dataset* d) {
  for (const Msg& m : msgs) {
    const uint64_t p = m.off, end = m.off + m.size;
    if (m.type == 0x01) {                       // Dataspace
      if (m.size < 4) return f.fail("HDF5: short dataspace message");
      const unsigned ver = f.b[p], rank = f.b[p + 1];
      if (rank > 8) return f.fail("HDF5: dataset rank > 8");
      uint64_t q;
      if (ver == 1) q = p + 8;
      else if (ver == 2) { if (f.b[p + 3] == 2) return f.fail("HDF5: null dataspace"); q = p + 4; }
      else return f.fail("HDF5: unknown dataspace version");
      if (q + (uint64_t)rank * f.sl > end) return f.fail("HDF5: short dataspace message");
      d->rank = (int)rank;
      for (unsigned i = 0; i < rank; ++i) d->dim[i] = (long long)f.rd(q + (uint64_t)i * f.sl, f.sl);
      d->has_space = true;
    } else if (m.type == 0x03) {                // Datatype
      if (m.size < 8) return f.fail("HDF5: short datatype message");
      d->tclass = f.b[p] & 0x0f;
      const unsigned bits0 = f.b[p + 1];
      d->tsize = (int)f.rd(p + 4, 4);
      d->big = bits0 & 1;
      if (d->tclass == 0) {
        d->tsigned = (bits0 & 0x08) != 0;
        if (d->tsize != 1 && d->tsize != 2 && d->tsize != 4 && d->tsize != 8) return f.fail("HDF5: integer dataset of unsupported size");
      } else if (d->tclass == 1) {
        if (m.size < 20) return f.fail("HDF5: short floating-point datatype");
        const unsigned prec = (unsigned)f.rd(p + 10, 2), eloc = f.b[p + 12], esz = f.b[p + 13], msz = f.b[p + 15];
        const long long bias = (long long)f.rd(p + 16, 4);
        const bool f32 = d->tsize == 4 && prec == 32 && eloc == 23 && esz == 8 && msz == 23 && bias == 127;
        const bool f64 = d->tsize == 8 && prec == 64 && eloc == 52 && esz == 11 && msz == 52 && bias == 1023;
        if (!(f32 || f64) || (bits0 & 0x40)) return f.fail("HDF5: floating-point dataset that is not IEEE binary32 / binary64");
      }
      d->has_type = true;
    } else if (m.type == 0x08) {                // Data layout
      if (m.size < 2) return f.fail("HDF5: short layout message");
      const unsigned ver = f.b[p];
      if (ver == 3 || ver == 4) {                // (version 4 -- libver=latest -- keeps the compact and contiguous forms of version 3)
        d->layout = f.b[p + 1];
        if (ver == 4 && d->layout >= 2) return f.fail("HDF5: version-4 chunk indexes / virtual layouts (libver=latest) are not supported");
        if (d->layout == 0) {
          if (p + 4 > end) return f.fail("HDF5: short layout message");
          d->size = f.rd(p + 2, 2); d->addr = p + 4;
          if (d->addr + d->size > end) return f.fail("HDF5: compact data overruns its message");
        } else if (d->layout == 1) {
          if (p + 2 + f.so + f.sl > end) return f.fail("HDF5: short layout message");
          d->addr = f.addr(p + 2); d->size = f.rd(p + 2 + f.so, f.sl);
        } else if (d->layout == 2) {
          d->cdims = f.b[p + 2];
          if (d->cdims < 2 || d->cdims > 9 || p + 3 + f.so + 4ull * d->cdims > end) return f.fail("HDF5: bad chunked layout message");
          d->addr = f.addr(p + 3);
          for (int i = 0; i < d->cdims; ++i) d->chunk[i] = f.rd(p + 3 + f.so + 4ull * i, 4);
        } else return f.fail("HDF5: unknown layout class");
      } else if (ver == 1 || ver == 2) {
        if (m.size < 8) return f.fail("HDF5: short layout message");
        const int nd = f.b[p + 1];
        d->layout = f.b[p + 2];
        uint64_t q = p + 8;
        if (d->layout != 0) { if (q + f.so > end) return f.fail("HDF5: short layout message"); d->addr = f.addr(q); q += f.so; }
        if (nd > 9 || q + 4ull * nd > end) return f.fail("HDF5: short layout message");
        uint64_t dims[9];
        for (int i = 0; i < nd; ++i) dims[i] = f.rd(q + 4ull * i, 4);
        q += 4ull * nd;
        if (d->layout == 2) {
          if (q + 4 > end || nd > 8) return f.fail("HDF5: short layout message");
          d->cdims = nd + 1;
          for (int i = 0; i < nd; ++i) d->chunk[i] = dims[i];
          d->chunk[nd] = f.rd(q, 4);
        } else if (d->layout == 0) {
          if (q + 4 > end) return f.fail("HDF5: short layout message");
          d->size = f.rd(q, 4); d->addr = q + 4;
          if (d->addr + d->size > end) return f.fail("HDF5: compact data overruns its message");
        } else {
          d->size = UNDEF;                      // contiguous, size implied by the dataspace
        }
      } else {
        return f.fail("HDF5: unknown data layout message version " + std::to_string(ver));
      }
      d->has_layout = true;
    } else if (m.type == 0x0b) {                // Filter pipeline
      if (m.size < 2) return f.fail("HDF5: short filter pipeline message");
      const unsigned ver = f.b[p], nf = f.b[p + 1];
      uint64_t q = p + (ver == 1 ? 8 : 2);
      if (ver != 1 && ver != 2) return f.fail("HDF5: unknown filter pipeline version");
      for (unsigned i = 0; i < nf; ++i) {
        if (q + 8 > end + 2) return f.fail("HDF5: short filter pipeline message");
        Dataset::Filter fl;
        fl.id = (unsigned)f.rd(q, 2); q += 2;
        uint64_t nlen = 0;
        if (ver == 1 || fl.id >= 256) { nlen = f.rd(q, 2); q += 2; }
        q += 2;                                 // flags
        const unsigned ncd = (unsigned)f.rd(q, 2); q += 2;
        if (ver == 1) nlen = (nlen + 7) & ~7ull;
        q += nlen;
        if (q + 4ull * ncd > end) return f.fail("HDF5: short filter pipeline message");
        for (unsigned k = 0; k < ncd; ++k) fl.cd.push_back((uint32_t)f.rd(q + 4ull * k, 4));
        q += 4ull * ncd;
        if (ver == 1 && (ncd & 1)) q += 4;
        d->filters.push_back(fl);
      }
    }
  }
  return true;
}

void walk(File& f, uint64_t header, const std::string& path, int depth, std::vector<std::pair<std::string, uint64_t>>* found,
          std::vector<uint64_t>* stack) {
  if (!f.err.empty()) return;
  if (depth > 16 || std::find(stack->begin(), stack->end(), header) != stack->end()) return;   // hard-link cycles
  std::vector<Msg> msgs;
  if (!header_messages(f, header, &msgs)) return;
  std::vector<Link> links;
  bool is_group = false;
  if (!group_links(f, msgs, &links, &is_group)) return;
  if (!is_group) {
    bool space = false, layout = false;
    for (const Msg& m : msgs) { space |= m.type == 0x01; layout |= m.type == 0x08; }
    if (space && layout) found->push_back({path, header});
    return;
  }
  stack->push_back(header);
  for (const Link& l : links) {
    if (l.header == UNDEF) continue;
    walk(f, l.header, path + "/" + l.name, depth + 1, found, stack);
  }
  stack->pop_back();
}

// finds the superblock (offset 0 or 512, 1024, ... behind a user block), sets the offset / length sizes and the base address and
// returns the root group's object header address
bool open_file(File& f, uint64_t* root) {
  static const unsigned char sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
  uint64_t at = 0;
  for (;; at = at ? at * 2 : 512) {
    if (!f.ok(at, 16)) return f.fail("not an HDF5 file (no superblock signature)");
    if (std::memcmp(f.b + at, sig, 8) == 0) break;
  }
  const unsigned ver = f.b[at + 8];
  if (ver <= 1) {
    f.so = f.b[at + 13]; f.sl = f.b[at + 14];
  } else if (ver <= 3) {
    f.so = f.b[at + 9]; f.sl = f.b[at + 10];
  } else {
    return f.fail("HDF5: unknown superblock version");
  }
  if ((f.so != 4 && f.so != 8) || (f.sl != 4 && f.sl != 8)) return f.fail("HDF5: unsupported offset / length size");
  if (ver <= 1) {
    uint64_t p = at + 24 + (ver == 1 ? 4 : 0);
    if (!f.ok(p, 6ull * f.so)) return f.fail("HDF5: truncated superblock");
    f.base = f.rd(p, f.so);                     // every address in the file is relative to the base address
    p += 4ull * f.so;                           // base, free-space info, end of file, driver info; then the root symbol-table entry
    *root = f.addr(p + f.so);
  } else {
    if (!f.ok(at + 12, 4ull * f.so)) return f.fail("HDF5: truncated superblock");
    f.base = f.rd(at + 12, f.so);
    *root = f.addr(at + 12 + 3ull * f.so);
  }
  if (*root == UNDEF) return f.fail("HDF5: no root group");
  return true;
}

double load_elem(const unsigned char* p, const Dataset& d) {
  unsigned char t[8];
  for (int i = 0; i < d.tsize; ++i) t[i] = d.big ? p[d.tsize - 1 - i] : p[i];
  if (d.tclass == 1) {
    if (d.tsize == 4) { float v; std::memcpy(&v, t, 4); return v; }
    double v; std::memcpy(&v, t, 8); return v;
  }
  uint64_t u = 0;
  for (int i = 0; i < d.tsize; ++i) u |= (uint64_t)t[i] << (8 * i);
  if (d.tsigned) {
    const int sh = 64 - 8 * d.tsize;
    return (double)((int64_t)(u << sh) >> sh);
  }
  return (double)u;
}

void convert(const unsigned char* src, const Dataset& d, float* dst, size_t n) {
  if (d.tclass == 1 && d.tsize == 4 && !d.big) { std::memcpy(dst, src, n * 4); return; }      // the Caffe case: bit copy
  for (size_t i = 0; i < n; ++i) dst[i] = (float)load_elem(src + i * d.tsize, d);
}

struct ChunkRef { uint64_t addr; uint32_t size, mask; uint64_t off[8]; };

bool btree_chunks(File& f, uint64_t node, int cdims, std::vector<ChunkRef>* out, int depth) {
  if (node == UNDEF) return true;               // no chunk was ever written
  if (depth > 32 || !f.ok(node, 8 + 2 * f.so)) return f.fail("HDF5: chunk B-tree node outside the file");
  if (std::memcmp(f.b + node, "TREE", 4) != 0 || f.b[node + 4] != 1) return f.fail("HDF5: bad chunk B-tree node");
  const unsigned level = f.b[node + 5], used = (unsigned)f.rd(node + 6, 2);
  const uint64_t ksz = 8 + 8ull * cdims;
  uint64_t p = node + 8 + 2 * f.so;
  if (!f.ok(p, used * (ksz + f.so) + ksz)) return f.fail("HDF5: chunk B-tree node outside the file");
  for (unsigned i = 0; i < used; ++i) {
    ChunkRef c;
    c.size = (uint32_t)f.rd(p, 4); c.mask = (uint32_t)f.rd(p + 4, 4);
    for (int k = 0; k < cdims - 1; ++k) c.off[k] = f.rd(p + 8 + 8ull * k, 8);
    c.addr = f.addr(p + ksz);
    p += ksz + f.so;
    if (level == 0) out->push_back(c);
    else if (!btree_chunks(f, c.addr, cdims, out, depth + 1)) return false;
  }
  return true;
}

bool undo_filters(File& f, const Dataset& d, const ChunkRef& c, size_t want, std::vector<unsigned char>* buf) {
  buf->assign(f.b + c.addr, f.b + c.addr + c.size);
  for (int i = (int)d.filters.size() - 1; i >= 0; --i) {
    if (c.mask & (1u << i)) continue;
    const Dataset::Filter& fl = d.filters[i];
    if (fl.id == 1) {                           // deflate
      std::vector<unsigned char> outb(want);
      uLongf n = (uLongf)want;
      const int rc = uncompress(outb.data(), &n, buf->data(), (uLong)buf->size());
      if (rc != Z_OK) return f.fail("HDF5: inflate of a chunk failed");
      outb.resize(n);
      buf->swap(outb);
    } else if (fl.id == 2) {                    // shuffle: byte planes back to elements
      const size_t es = fl.cd.empty() ? (size_t)d.tsize : fl.cd[0];
      if (es > 1) {
        const size_t n = buf->size() / es;
        std::vector<unsigned char> outb(buf->size());
        for (size_t k = 0; k < es; ++k)
          for (size_t j = 0; j < n; ++j) outb[j * es + k] = (*buf)[k * n + j];
        for (size_t j = n * es; j < buf->size(); ++j) outb[j] = (*buf)[j];
        buf->swap(outb);
      }
    } else if (fl.id == 3) {                    // fletcher32: checksum trails the data
      if (buf->size() >= 4) buf->resize(buf->size() - 4);
    } else {
      return f.fail("HDF5: filter " + std::to_string(fl.id) + " is not supported (deflate, shuffle, fletcher32 are)");
    }
  }
  return true;
}

bool read_dataset(File& f, const Dataset& d, float* dst, size_t count) {
  const size_t es = (size_t)d.tsize;
  if (d.layout == 0 || d.layout == 1) {
    if (!d.filters.empty() && d.layout == 1) return f.fail("HDF5: filters on a contiguous dataset");
    if (d.addr == UNDEF) { std::fill(dst, dst + count, 0.0f); return true; }     // never written: the fill value (0)
    if (!f.ok(d.addr, count * es) || (d.size != UNDEF && d.size < count * es)) return f.fail("HDF5: dataset payload outside the file");
    convert(f.b + d.addr, d, dst, count);
    return true;
  }
  const int rank = d.rank;
  if (d.cdims != rank + 1 || d.chunk[rank] != es) return f.fail("HDF5: chunk shape does not match the dataset");
  std::vector<ChunkRef> chunks;
  if (!btree_chunks(f, d.addr, d.cdims, &chunks, 0)) return false;
  std::fill(dst, dst + count, 0.0f);
  size_t celems = 1;
  for (int i = 0; i < rank; ++i) { if (d.chunk[i] == 0) return f.fail("HDF5: empty chunk shape"); celems *= (size_t)d.chunk[i]; }
  std::vector<unsigned char> buf;
  std::vector<float> row;
  for (const ChunkRef& c : chunks) {
    if (!f.ok(c.addr, c.size)) return f.fail("HDF5: chunk outside the file");
    if (d.filters.empty()) buf.assign(f.b + c.addr, f.b + c.addr + c.size);
    else if (!undo_filters(f, d, c, celems * es, &buf)) return false;
    if (buf.size() < celems * es) return f.fail("HDF5: short chunk");
    // copy the part of the chunk that lies inside the dataset, innermost rows at a time
    const size_t inner = rank ? (size_t)d.chunk[rank - 1] : 1;
    const size_t rows = celems / inner;
    for (size_t r = 0; r < rows; ++r) {
      size_t rem = r, doff = 0, stride = 1;
      bool inside = true;
      // coordinates of this row inside the chunk (all axes but the last), dataset offset built back to front
      long long coord[8];
      for (int a = rank - 2; a >= 0; --a) { coord[a] = (long long)(rem % d.chunk[a]); rem /= d.chunk[a]; }
      stride = rank ? (size_t)d.dim[rank - 1] : 1;
      for (int a = rank - 2; a >= 0; --a) {
        const long long g = (long long)c.off[a] + coord[a];
        if (g >= d.dim[a]) { inside = false; break; }
        doff += (size_t)g * stride;
        stride *= (size_t)d.dim[a];
      }
      if (!inside) continue;
      const long long x0 = rank ? (long long)c.off[rank - 1] : 0;
      const long long lim = rank ? d.dim[rank - 1] : 1;
      if (x0 >= lim) continue;
      const size_t n = (size_t)std::min<long long>((long long)inner, lim - x0);
      convert(buf.data() + r * inner * es, d, dst + doff + (size_t)x0, n);
    }
  }
  return true;
}

int set_error(const File& f) { return fn2::fail(FN2_ERR_INVALID_ARG, "%s", f.err.empty() ? "HDF5: malformed file" : f.err.c_str()); }

}  // namespace

extern "C" {

FN2_API int fn2_hdf5_index(const void* buf, size_t len, fn2_hdf5_entry* entries, int max_entries, int* num_entries) {
  if (!buf || !num_entries || max_entries < 0 || (max_entries > 0 && !entries)) return fn2::fail(FN2_ERR_INVALID_ARG, "fn2_hdf5_index: null argument");
  File f{(const unsigned char*)buf, len};
  uint64_t root;
  if (!open_file(f, &root)) return set_error(f);
  std::vector<std::pair<std::string, uint64_t>> found;
  std::vector<uint64_t> stack;
  walk(f, root, "", 0, &found, &stack);
  if (!f.err.empty()) return set_error(f);
  int n = 0;
  for (const auto& it : found) {
    std::vector<Msg> msgs;
    Dataset d;
    if (!header_messages(f, it.second, &msgs) || !parse_dataset(f, msgs, &d)) return set_error(f);
    if (!d.has_space || !d.has_type || !d.has_layout) continue;
    if (n < max_entries) {
      fn2_hdf5_entry* e = &entries[n];
      std::memset(e, 0, sizeof(*e));
      if (it.first.size() >= sizeof(e->path)) return fn2::fail(FN2_ERR_INVALID_ARG, "HDF5: dataset path longer than 255 bytes");
      std::memcpy(e->path, it.first.c_str(), it.first.size());
      e->num_axes = d.rank;
      e->count = 1;
      for (int i = 0; i < d.rank; ++i) { e->dim[i] = d.dim[i]; e->count *= (size_t)d.dim[i]; }
      e->type_class = d.tclass; e->type_size = d.tsize; e->type_signed = d.tsigned; e->big_endian = d.big;
      e->layout = d.layout; e->num_filters = (int)d.filters.size();
      e->header_off = (size_t)it.second;
    }
    ++n;
  }
  *num_entries = n;
  return FN2_OK;
}

FN2_API int fn2_hdf5_read_float(const void* buf, size_t len, const fn2_hdf5_entry* entry, float* dst, size_t dst_floats) {
  if (!buf || !entry || (!dst && dst_floats)) return fn2::fail(FN2_ERR_INVALID_ARG, "fn2_hdf5_read_float: null argument");
  File f{(const unsigned char*)buf, len};
  uint64_t root;
  if (!open_file(f, &root)) return set_error(f);
  std::vector<Msg> msgs;
  Dataset d;
  if (!header_messages(f, entry->header_off, &msgs) || !parse_dataset(f, msgs, &d)) return set_error(f);
  if (!d.has_space || !d.has_type || !d.has_layout) return fn2::fail(FN2_ERR_INVALID_ARG, "fn2_hdf5_read_float: the entry is not a dataset");
  // hdf5_load_nd_dataset_helper (util/hdf5.cpp:26-52): H5T_FLOAT and H5T_INTEGER are read, every other class is LOG(FATAL)
  if (d.tclass != 0 && d.tclass != 1) return fn2::fail(FN2_ERR_INVALID_ARG, "Unsupported datatype class (only H5T_FLOAT and H5T_INTEGER are read)");
  size_t count = 1;
  for (int i = 0; i < d.rank; ++i) count *= (size_t)d.dim[i];
  if (count != dst_floats) return fn2::fail(FN2_ERR_INVALID_ARG, "fn2_hdf5_read_float: destination size differs from the dataset's element count");
  if (!read_dataset(f, d, dst, count)) return set_error(f);
  return FN2_OK;
}

}  // extern "C"
