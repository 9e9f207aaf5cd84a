// A PopulationDB the way a parser would deliver it (shared Variant objects per record/alt/phase, a few genomes holding
// private copies, repeated records) through gpu::flattenPopulation with 1 thread and with many: the two results must be
// identical.  Built and run by tests/tools/sanitize_host.sh under ASan/UBSan and under TSan (the flattener's discovery
// walk and its packing loop are both multi-threaded); no GPU, no oracle.
#include <cstdio>
#include <map>
#include <random>

#include "kgx_flatten.h"

namespace kgl = kellerberrin::genome;
namespace gpu = kellerberrin::genome::analysis::gpu;

int main() {
  constexpr size_t G = 203, L = 900;
  std::mt19937_64 rng(12345);
  auto population = std::make_shared<kgl::PopulationDB>("check", kgl::DataSourceEnum::Genome1000);
  std::vector<kgl::GenomeId_t> ids;
  for (size_t g = 0; g < G; ++g) ids.push_back("HG" + std::to_string(100000 + g));
  const char bases[] = "ACGT";
  for (size_t l = 0; l < L; ++l) {
    const kgl::ContigOffset_t offset = 1000 + 7 * (l / 2);                 // two records per offset: compound offsets
    const size_t n_alt = 1 + rng() % 3;
    auto info = std::make_shared<kgl::InfoRecord>();
    std::vector<float> af(n_alt);
    for (auto& f : af) f = static_cast<float>(rng() % 1000) / 1000.0f;
    info->float_fields["AF"] = af;
    for (size_t a = 0; a < n_alt; ++a) {
      std::string alt(1, bases[(l + a + 1) % 4]);
      if (a == 2) alt += "T";
      std::shared_ptr<const kgl::Variant> phase[2];
      for (int p = 0; p < 2; ++p) {
        kgl::VariantEvidence evidence(l, kgl::DataSourceEnum::Genome1000, true, info, static_cast<uint32_t>(a), static_cast<uint32_t>(n_alt));
        phase[p] = std::make_shared<const kgl::Variant>("chr1", offset, p ? kgl::VariantPhase::DIPLOID_PHASE_B : kgl::VariantPhase::DIPLOID_PHASE_A,
                                                        "", kgl::DNA5SequenceLinear(std::string(1, bases[l % 4])), kgl::DNA5SequenceLinear(alt), evidence);
      }
      for (size_t g = 0; g < G; ++g) {
        const unsigned draw = rng() % 16;
        if (draw < 3) (void)population->addVariant(phase[draw & 1], {ids[g]});
        else if (draw == 3) { (void)population->addVariant(phase[0], {ids[g]}); (void)population->addVariant(phase[1], {ids[g]}); }
        else if (draw == 4 && g % 17 == 0) {                                // more than two copies: one object three times
          for (int k = 0; k < 3; ++k) (void)population->addVariant(phase[0], {ids[g]});
        }
      }
    }
  }
  const gpu::FlatPopulation one = gpu::flattenPopulation(*population, 1);
  int failures = 0;
  {
    // the phase plane against a direct count: a cell's bit is set exactly where the genome holds the row's variant on both phases
    size_t set_bits = 0, expected = 0;
    for (const uint8_t byte : one.phase_plane) set_bits += static_cast<size_t>(__builtin_popcount(byte));
    for (const auto& [genome_id, genome_ptr] : population->getMap())
      for (const auto& [contig_id, contig_ptr] : genome_ptr->getMap())
        for (const auto& [offset, offset_ptr] : contig_ptr->getMap()) {
          std::map<std::string, unsigned> phases_of;
          for (const auto& v : offset_ptr->getVariantArray()) phases_of[v->HGVS()] |= v->phaseId() == kgl::VariantPhase::DIPLOID_PHASE_A ? 1u : 2u;
          for (const auto& [hgvs, mask] : phases_of) expected += mask == 3u ? 1 : 0;
        }
    std::printf("phase plane: %zu cells on both phases, %zu expected\n", set_bits, expected);
    failures += (set_bits == expected && expected > 0) ? 0 : 1;
  }
  for (const size_t threads : {2u, 7u, 64u}) {
    const gpu::FlatPopulation many = gpu::flattenPopulation(*population, threads);
    bool same = many.packed == one.packed && many.phase_plane == one.phase_plane && many.genome_ids == one.genome_ids && many.rows.size() == one.rows.size() &&
                many.variant_objects == one.variant_objects && many.non_diploid.size() == one.non_diploid.size() && many.primary_rows == one.primary_rows;
    for (size_t r = 0; same && r < one.rows.size(); ++r)
      same = many.rows[r].hgvs == one.rows[r].hgvs && many.rows[r].variant == one.rows[r].variant && many.rows[r].split_of == one.rows[r].split_of &&
             ((many.rows[r].info_af == one.rows[r].info_af) || (many.rows[r].info_af != many.rows[r].info_af && one.rows[r].info_af != one.rows[r].info_af));
    std::printf("flattenPopulation with %zu threads: %s (%zu rows, %zu genomes, %llu Variant objects, %zu cells over two copies)\n", threads,
                same ? "identical to 1 thread" : "DIFFERENT", many.rows.size(), many.genome_ids.size(), (unsigned long long)many.variant_objects, many.non_diploid.size());
    failures += same ? 0 : 1;
  }
  return failures;
}
