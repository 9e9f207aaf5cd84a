#include "kgx_vcf_io.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <thread>
#include <vector>

namespace kellerberrin::genome::analysis::gpu {

namespace {

struct Block {
  size_t data_begin;        // first byte of the raw deflate stream
  size_t data_size;
  size_t out_begin;
  uint32_t out_size;        // ISIZE of the trailer
  uint32_t crc;             // CRC32 of the trailer
};

uint16_t le16(const unsigned char* p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }
uint32_t le32(const unsigned char* p) { return static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) | (static_cast<uint32_t>(p[2]) << 16) | (static_cast<uint32_t>(p[3]) << 24); }

// Index the members of a block-gzip file.  false if some member is not a BGZF block (then the file is plain gzip).
bool indexBlocks(const std::string& raw, std::vector<Block>& blocks, size_t& total) {
  const auto* p = reinterpret_cast<const unsigned char*>(raw.data());
  size_t at = 0;
  total = 0;
  while (at < raw.size()) {
    if (raw.size() - at < 18) return false;
    if (p[at] != 31 || p[at + 1] != 139 || p[at + 2] != 8 || !(p[at + 3] & 4)) return false;     // gzip, deflate, FEXTRA
    const size_t xlen = le16(p + at + 10);
    if (raw.size() - at < 12 + xlen + 8) return false;
    size_t block_size = 0;
    for (size_t x = at + 12; x + 4 <= at + 12 + xlen;) {                                            // extra subfields
      const size_t len = le16(p + x + 2);
      if (p[x] == 'B' && p[x + 1] == 'C' && len == 2 && x + 6 <= at + 12 + xlen) block_size = static_cast<size_t>(le16(p + x + 4)) + 1;
      x += 4 + len;
    }
    if (block_size == 0 || block_size < 12 + xlen + 8 || raw.size() - at < block_size) return false;
    if (p[at + 3] & ~4) return false;                                                               // other header fields: not bgzip's output
    Block b;
    b.data_begin = at + 12 + xlen;
    b.data_size = block_size - (12 + xlen) - 8;
    b.crc = le32(p + at + block_size - 8);
    b.out_size = le32(p + at + block_size - 4);
    if (b.out_size > 65536) return false;
    b.out_begin = total;
    total += b.out_size;
    blocks.push_back(b);
    at += block_size;
  }
  return true;
}

bool inflateBlock(const std::string& raw, const Block& b, char* out) {
  if (b.out_size == 0) return true;                                                                 // the EOF marker block
  z_stream zs;
  std::memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, -15) != Z_OK) return false;
  zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(raw.data() + b.data_begin));
  zs.avail_in = static_cast<uInt>(b.data_size);
  zs.next_out = reinterpret_cast<Bytef*>(out);
  zs.avail_out = b.out_size;
  const int rc = inflate(&zs, Z_FINISH);
  const bool ok = rc == Z_STREAM_END && zs.total_out == b.out_size;
  inflateEnd(&zs);
  return ok && crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef*>(out), b.out_size) == b.crc;
}

// Plain gzip: one thread, members concatenated.
bool inflateGzip(const std::string& raw, std::string& text) {
  z_stream zs;
  std::memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
  zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(raw.data()));
  zs.avail_in = static_cast<uInt>(std::min<size_t>(raw.size(), 1u << 30));
  size_t consumed_base = 0;
  std::vector<char> chunk(1 << 20);
  text.clear();
  for (;;) {
    zs.next_out = reinterpret_cast<Bytef*>(chunk.data());
    zs.avail_out = static_cast<uInt>(chunk.size());
    const int rc = inflate(&zs, Z_NO_FLUSH);
    text.append(chunk.data(), chunk.size() - zs.avail_out);
    if (rc == Z_STREAM_END) {
      const size_t used = consumed_base + (reinterpret_cast<const char*>(zs.next_in) - (raw.data() + consumed_base));
      if (used >= raw.size()) break;
      consumed_base = used;                                   // next member
      if (inflateReset(&zs) != Z_OK) { inflateEnd(&zs); return false; }
      zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(raw.data() + consumed_base));
      zs.avail_in = static_cast<uInt>(std::min<size_t>(raw.size() - consumed_base, 1u << 30));
      continue;
    }
    if (rc != Z_OK) { inflateEnd(&zs); return false; }
    if (zs.avail_in == 0) {
      const size_t used = reinterpret_cast<const char*>(zs.next_in) - raw.data();
      if (used >= raw.size()) { inflateEnd(&zs); return false; }   // truncated
      zs.avail_in = static_cast<uInt>(std::min<size_t>(raw.size() - used, 1u << 30));
    }
  }
  inflateEnd(&zs);
  return true;
}

}  // namespace

bool readVcfText(const std::string& file_name, std::string& text, std::string& error, size_t threads) {
  std::ifstream in(file_name, std::ios::binary);
  if (!in.good()) { error = "cannot open file: " + file_name; return false; }
  std::string raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  if (raw.size() < 2 || static_cast<unsigned char>(raw[0]) != 31 || static_cast<unsigned char>(raw[1]) != 139) {
    text = std::move(raw);
    return true;
  }
  std::vector<Block> blocks;
  size_t total = 0;
  if (!indexBlocks(raw, blocks, total)) {
    if (!inflateGzip(raw, text)) { error = "not a valid gzip file: " + file_name; return false; }
    return true;
  }
  text.assign(total, '\0');
  if (threads == 0) threads = std::max<size_t>(std::thread::hardware_concurrency(), 2) - 1;
  std::atomic<size_t> next{0};
  std::atomic<bool> failed{false};
  auto worker = [&]() {
    for (size_t b = next.fetch_add(1); b < blocks.size() && !failed.load(); b = next.fetch_add(1))
      if (!inflateBlock(raw, blocks[b], &text[blocks[b].out_begin])) failed.store(true);
  };
  const size_t n = std::max<size_t>(1, std::min(threads, blocks.size()));
  std::vector<std::thread> pool;
  for (size_t t = 1; t < n; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
  if (failed.load()) { error = "block gzip file fails its size / CRC check: " + file_name; return false; }
  return true;
}

// ---- bounded pieces ---------------------------------------------------------------------------------------------------

struct VcfChunkReader::State {
  enum class Kind { Plain, Gzip, BlockGzip } kind{Kind::Plain};
  std::ifstream in;
  std::string file_name;
  size_t threads{1};
  size_t chunk_bytes{0};
  bool source_done{false};
  std::string carry;            // the incomplete last line of the previous piece
  std::string raw;              // compressed bytes read but not yet consumed (block gzip: a partial block; gzip: input window)
  z_stream zs;                  // plain gzip
  bool zs_open{false};
  size_t zs_in_at{0};           // consumed prefix of raw

  // Append up to `want` bytes of the file to raw; false when nothing was left.
  bool readMore(size_t want) {
    const size_t had = raw.size();
    raw.resize(had + want);
    in.read(&raw[had], static_cast<std::streamsize>(want));
    const size_t got = static_cast<size_t>(in.gcount());
    raw.resize(had + got);
    return got > 0;
  }
};

VcfChunkReader::VcfChunkReader() : state_(new State()) {}
VcfChunkReader::~VcfChunkReader() {
  if (state_->zs_open) inflateEnd(&state_->zs);
  delete state_;
}

bool VcfChunkReader::open(const std::string& file_name, std::string& error, size_t threads, size_t chunk_bytes) {
  State& st = *state_;
  st.in.open(file_name, std::ios::binary);
  if (!st.in.good()) { error = "cannot open file: " + file_name; return false; }
  st.file_name = file_name;
  st.threads = threads ? threads : std::max<size_t>(std::thread::hardware_concurrency(), 2) - 1;
  st.chunk_bytes = std::max<size_t>(chunk_bytes, 1);
  st.readMore(1 << 16);
  const auto* p = reinterpret_cast<const unsigned char*>(st.raw.data());
  if (st.raw.size() < 2 || p[0] != 31 || p[1] != 139) {
    st.kind = State::Kind::Plain;
  } else {
    // a first member with the "BC" subfield is block gzip; every later member must be one too
    bool bgzf = st.raw.size() >= 18 && p[2] == 8 && p[3] == 4;
    if (bgzf) {
      const size_t xlen = le16(p + 10);
      bgzf = false;
      for (size_t x = 12; x + 4 <= 12 + xlen && x + 4 <= st.raw.size();) {
        const size_t len = le16(p + x + 2);
        if (p[x] == 'B' && p[x + 1] == 'C' && len == 2) bgzf = true;
        x += 4 + len;
      }
    }
    st.kind = bgzf ? State::Kind::BlockGzip : State::Kind::Gzip;
    if (!bgzf) {
      std::memset(&st.zs, 0, sizeof(st.zs));
      if (inflateInit2(&st.zs, 15 + 16) != Z_OK) { error = "zlib initialisation failed"; return false; }
      st.zs_open = true;
    }
  }
  return true;
}

bool VcfChunkReader::next(std::string& text, std::string& error) {
  State& st = *state_;
  error.clear();
  text = std::move(st.carry);
  st.carry.clear();
  // fill: append about chunk_bytes of text
  while (!st.source_done && text.size() < st.chunk_bytes) {
    if (st.kind == State::Kind::Plain) {
      if (!st.raw.empty()) { text += st.raw; st.raw.clear(); continue; }
      if (!st.readMore(std::min<size_t>(st.chunk_bytes, size_t{64} << 20))) st.source_done = true;
    } else if (st.kind == State::Kind::BlockGzip) {
      // whole blocks held in raw -> text, in parallel; a partial block stays for the next read
      std::vector<Block> blocks;
      const auto* p = reinterpret_cast<const unsigned char*>(st.raw.data());
      size_t at = 0, total = 0;
      while (st.raw.size() - at >= 18) {
        if (p[at] != 31 || p[at + 1] != 139 || p[at + 2] != 8 || p[at + 3] != 4) { error = "not a block gzip member in: " + st.file_name; return false; }
        const size_t xlen = le16(p + at + 10);
        if (st.raw.size() - at < 12 + xlen) break;
        size_t block_size = 0;
        for (size_t x = at + 12; x + 4 <= at + 12 + xlen;) {
          const size_t len = le16(p + x + 2);
          if (p[x] == 'B' && p[x + 1] == 'C' && len == 2 && x + 6 <= at + 12 + xlen) block_size = static_cast<size_t>(le16(p + x + 4)) + 1;
          x += 4 + len;
        }
        if (block_size == 0 || block_size < 12 + xlen + 8) { error = "not a block gzip member in: " + st.file_name; return false; }
        if (st.raw.size() - at < block_size) break;
        Block b;
        b.data_begin = at + 12 + xlen;
        b.data_size = block_size - (12 + xlen) - 8;
        b.crc = le32(p + at + block_size - 8);
        b.out_size = le32(p + at + block_size - 4);
        if (b.out_size > 65536) { error = "block gzip member larger than 64 KiB in: " + st.file_name; return false; }
        b.out_begin = total;
        total += b.out_size;
        blocks.push_back(b);
        at += block_size;
        if (text.size() + total >= st.chunk_bytes) break;
      }
      if (!blocks.empty()) {
        const size_t base = text.size();
        text.resize(base + total);
        std::atomic<size_t> next_block{0};
        std::atomic<bool> failed{false};
        auto worker = [&]() {
          for (size_t b = next_block.fetch_add(1); b < blocks.size() && !failed.load(); b = next_block.fetch_add(1))
            if (!inflateBlock(st.raw, blocks[b], &text[base + blocks[b].out_begin])) failed.store(true);
        };
        const size_t n = std::max<size_t>(1, std::min(st.threads, blocks.size()));
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n; ++t) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
        if (failed.load()) { error = "block gzip file fails its size / CRC check: " + st.file_name; return false; }
        st.raw.erase(0, at);
        continue;
      }
      if (!st.readMore(std::max<size_t>(size_t{1} << 20, std::min<size_t>(st.chunk_bytes / 4, size_t{64} << 20)))) {
        if (!st.raw.empty()) { error = "truncated block gzip file: " + st.file_name; return false; }
        st.source_done = true;
      }
    } else {
      // plain gzip: members concatenated, one stream
      if (st.zs_in_at >= st.raw.size()) {
        st.raw.clear();
        st.zs_in_at = 0;
        if (!st.readMore(size_t{4} << 20)) {
          error = "truncated gzip file: " + st.file_name;      // the stream's end is seen below, before the input runs dry
          return false;
        }
      }
      const size_t base = text.size();
      const size_t room = std::max<size_t>(size_t{1} << 20, std::min<size_t>(st.chunk_bytes - std::min(st.chunk_bytes, base), size_t{64} << 20));
      text.resize(base + room);
      st.zs.next_in = reinterpret_cast<Bytef*>(&st.raw[st.zs_in_at]);
      st.zs.avail_in = static_cast<uInt>(st.raw.size() - st.zs_in_at);
      st.zs.next_out = reinterpret_cast<Bytef*>(&text[base]);
      st.zs.avail_out = static_cast<uInt>(room);
      const int rc = inflate(&st.zs, Z_NO_FLUSH);
      st.zs_in_at = st.raw.size() - st.zs.avail_in;
      text.resize(base + (room - st.zs.avail_out));
      if (rc == Z_STREAM_END) {
        if (st.zs_in_at >= st.raw.size()) {                       // more members?
          st.raw.clear();
          st.zs_in_at = 0;
          if (!st.readMore(size_t{4} << 20)) { st.source_done = true; continue; }
        }
        if (inflateReset(&st.zs) != Z_OK) { error = "zlib reset failed"; return false; }
      } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
        error = "not a valid gzip file: " + st.file_name;
        return false;
      }
    }
  }
  if (st.source_done) return !text.empty();
  // cut at the last line end; the rest opens the next piece
  const size_t cut = text.rfind('\n');
  if (cut == std::string::npos) {
    // one line longer than a piece: keep filling
    std::string more, err;
    st.carry = std::move(text);
    const size_t keep = st.chunk_bytes;
    st.chunk_bytes = st.carry.size() * 2;
    const bool got = next(more, err);
    st.chunk_bytes = keep;
    if (!err.empty()) { error = err; return false; }
    text = std::move(more);
    return got;
  }
  st.carry.assign(text, cut + 1, std::string::npos);
  text.resize(cut + 1);
  return true;
}

}  // namespace kellerberrin::genome::analysis::gpu
