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

}  // namespace kellerberrin::genome::analysis::gpu

#endif  // KGX_VCF_IO_H
