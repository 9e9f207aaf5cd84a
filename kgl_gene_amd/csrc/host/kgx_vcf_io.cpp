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

}  // namespace kellerberrin::genome::analysis::gpu
