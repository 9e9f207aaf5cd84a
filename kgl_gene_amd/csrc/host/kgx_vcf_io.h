// VCF file -> text in memory.  Plain text, gzip (RFC 1952, any number of members) and block gzip (.bgz / bgzip: gzip
// members of at most 64 KiB uncompressed, each carrying its own compressed size in a "BC" extra subfield) -- the format
// the large VCFs of the path come in.  The reference reads .bgz through a multi-threaded block pipeline
// (kel_io/kel_bzip_workflow.h:22-35, verifying each block); here the blocks are indexed in one pass over the headers and
// then inflated in parallel straight into their places in the output, each checked against its CRC32 and size.
#ifndef KGX_VCF_IO_H
#define KGX_VCF_IO_H

#include <string>

namespace kellerberrin::genome::analysis::gpu {

// threads == 0: hardware_concurrency() - 1.  Returns false with a message in error.
[[nodiscard]] bool readVcfText(const std::string& file_name, std::string& text, std::string& error, size_t threads = 0);

// The same file a bounded piece at a time, for VCFs whose text does not fit in memory (a 10 k-sample x 10 M-record
// file is ~400 GB of text for 25 GB of genotypes): every call to next() yields the next run of WHOLE lines, about
// chunk_bytes of text.  Plain files are read in slices; block gzip is read a batch of blocks at a time, the blocks
// inflated in parallel and checked like readVcfText's; plain gzip is inflated as a stream by one thread.
class VcfChunkReader {
 public:
  VcfChunkReader();
  ~VcfChunkReader();
  VcfChunkReader(const VcfChunkReader&) = delete;
  VcfChunkReader& operator=(const VcfChunkReader&) = delete;
  [[nodiscard]] bool open(const std::string& file_name, std::string& error, size_t threads = 0, size_t chunk_bytes = size_t{64} << 20);
  // false at the end of the file, or on an error (then error is not empty).  text is overwritten.
  [[nodiscard]] bool next(std::string& text, std::string& error);

 private:
  struct State;
  State* state_;
};

}  // namespace kellerberrin::genome::analysis::gpu

#endif  // KGX_VCF_IO_H
