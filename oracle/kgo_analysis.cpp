// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see kgo_core.h).
#include "kgo_analysis.h"

namespace kgo {

// ---- VariantDBVariant (kgl_variant_db_variant.cpp) ---------------------------------------------

void VariantDBVariant::createVariantDB(const std::shared_ptr<const PopulationDB>& population) {
  // Variant index = rank of HGVS() in a std::map<string> (lexicographic) (:17-30).
  auto unique_variant_map = population->uniqueVariants();
  size_t index = 0;
  for (auto& [hgvs, variant_ptr] : unique_variant_map) {
    auto [it, ok] = variant_index_.try_emplace(hgvs, std::pair<VariantPtr, size_t>{variant_ptr, index});
    if (!ok) continue;
    ++index;
  }
  const size_t variant_size = variant_index_.size();

  // Genome index = rank of genome id; one zeroed uint8 dosage vector per genome (:36-51).
  index = 0;
  for (auto& [genome_id, genome_ptr] : population->getMap()) {
    auto [it, ok] = genome_index_.try_emplace(genome_id, index);
    if (!ok) continue;
    genome_data_.emplace_back(genome_id, std::vector<uint8_t>(variant_size, 0));
    ++index;
  }

  // One pool task per genome; every Variant visit formats HGVS, finds it, ++cell (:65-107,117).
  population->processAll_MT([this](const std::shared_ptr<const GenomeDB>& genome_ptr, const VariantPtr& variant_ptr) {
    auto hgvs = variant_ptr->HGVS();
    auto vit = variant_index_.find(hgvs);
    if (vit == variant_index_.end()) return false;
    const size_t var_index = vit->second.second;
    auto git = genome_index_.find(genome_ptr->genomeId());
    if (git == genome_index_.end()) return false;
    auto& variant_array = genome_data_.at(git->second).second;
    ++(variant_array[var_index]);
    return true;
  });
}

AlleleSummmary VariantDBVariant::summaryByVariant(const VariantPtr& variant) const {
  AlleleSummmary s;
  auto it = variant_index_.find(variant->HGVS());
  if (it == variant_index_.end()) return s;
  const size_t variant_index = it->second.second;
  for (const auto& [genome, variant_vector] : genome_data_) {
    switch (variant_vector[variant_index]) {
      case 0: ++s.referenceHomozygous_; break;
      case 1: ++s.minorHeterozygous_; break;
      case 2: ++s.minorHomozygous_; break;
      default: break;   // non-diploid: warned, not counted (:158-161)
    }
  }
  if (genome_data_.size() != s.referenceHomozygous_ + s.minorHomozygous_ + s.minorHeterozygous_) ++warnings_;
  return s;
}

AlleleSummmary VariantDBVariant::summaryByGenome(const std::string& genome) const {
  AlleleSummmary s;
  auto it = genome_index_.find(genome);
  if (it == genome_index_.end()) return s;
  const auto& allele_vector = genome_data_[it->second].second;
  for (const auto allele_type : allele_vector) {
    switch (allele_type) {
      case 0: ++s.referenceHomozygous_; break;
      case 1: ++s.minorHeterozygous_; break;
      case 2: ++s.minorHomozygous_; break;
      default: break;
    }
  }
  if (variant_index_.size() != s.referenceHomozygous_ + s.minorHomozygous_ + s.minorHeterozygous_) ++warnings_;
  return s;
}

AlleleSummmary VariantDBVariant::populationSummary() const {
  AlleleSummmary s;
  for (const auto& [genome_id, allele_vector] : genome_data_) {
    for (const auto allele_type : allele_vector) {
      switch (allele_type) {
        case 0: ++s.referenceHomozygous_; break;
        case 1: ++s.minorHeterozygous_; break;
        case 2: ++s.minorHomozygous_; break;
        default: break;
      }
    }
  }
  if (variant_index_.size() * genome_data_.size() != s.referenceHomozygous_ + s.minorHomozygous_ + s.minorHeterozygous_)
    ++warnings_;
  return s;
}

// ---- P7FrequencyFilter / CalcFWS ---------------------------------------------------------------

bool p7FrequencyFilter(const Variant& v, double freq_cutoff) {
  // getTypedInfoData<vector<double>>("AF"): the float32 values widened to double; a missing field or a
  // missing value for this alt lets the variant through (:61-64).  Size mismatch -> false (:31-44).
  const RecordEvidence& ev = v.evidence();
  if (ev.af.size() != static_cast<size_t>(SUPER_POP_COUNT) * ev.alt_count || ev.info_af_size == -1) return true;   // field absent
  if (ev.info_af_size >= 0 && static_cast<uint32_t>(ev.info_af_size) != ev.alt_count) return false;               // vector size != alt count (:31-37)
  if (v.altVariantIndex() >= ev.alt_count) return false;
  const float f = ev.af[static_cast<size_t>(ALL) * ev.alt_count + v.altVariantIndex()];
  if (std::isnan(f)) return true;
  return static_cast<double>(f) >= freq_cutoff;
}

std::pair<double, double> CalcFWS::getFrequency(size_t bin) {
  switch (bin) {
    case 0: return {0.0, 0.05};
    case 1: return {0.05, 0.10};
    case 2: return {0.10, 0.15};
    case 3: return {0.15, 0.20};
    case 4: return {0.20, 0.25};
    case 5: return {0.25, 0.30};
    case 6: return {0.30, 0.35};
    case 7: return {0.35, 0.40};
    case 8: return {0.40, 0.45};
    case 9: return {0.45, 0.5};
    case 10: return {0.5, 1.0};
  }
  return {0.0, 0.0};
}

void CalcFWS::calcFwsStatistics(const std::shared_ptr<const PopulationDB>& population) {
  updateVariantFWSMap(population);
  for (size_t i = 0; i < FWS_FREQUENCY_ARRAY_SIZE; ++i) {
    const auto [lower_freq, upper_freq] = getFrequency(i);
    // AndFilter(P7Freq(lower), NotFilter(P7Freq(upper))) (:27-29)
    const double lo = lower_freq, hi = upper_freq;
    std::shared_ptr<const PopulationDB> freq_population = population->viewFilter(
        [lo, hi](const Variant& v) { return p7FrequencyFilter(v, lo) && !p7FrequencyFilter(v, hi); });
    updateGenomeFWSMap(freq_population, i);
  }
}

void CalcFWS::updateVariantFWSMap(const std::shared_ptr<const PopulationDB>& population) {
  VariantDBVariant variant_db_variant(population);
  for (const auto& [hgvs, variant_record] : variant_db_variant.variantMap()) {
    auto it = variant_fws_map_.find(hgvs);
    if (it == variant_fws_map_.end()) it = variant_fws_map_.try_emplace(hgvs, AlleleSummmary()).first;
    it->second += variant_db_variant.summaryByVariant(variant_record.first);
  }
}

void CalcFWS::updateGenomeFWSMap(const std::shared_ptr<const PopulationDB>& freq_population, size_t freq_bin) {
  VariantDBVariant variant_db_variant(freq_population);
  for (const auto& [genome_id, genome_ptr] : freq_population->getMap()) {
    auto it = genome_fws_map_.find(genome_id);
    if (it == genome_fws_map_.end()) it = genome_fws_map_.try_emplace(genome_id, FwsFrequencyArray()).first;
    it->second[freq_bin] += variant_db_variant.summaryByGenome(genome_id);
  }
}

// ---- HeteroHomoZygous --------------------------------------------------------------------------

void updateVariantAnalysisType(const OffsetDB& offset, VariantAnalysisType& rec) {
  if (offset.getVariantArray().empty()) return;
  for (const auto& v : offset.getVariantArray()) {
    ++rec.total_variants_;
    if (v->isSNP()) ++rec.snp_count_;
    else ++rec.indel_count_;
  }
  if (offset.getVariantArray().size() == 1) {
    ++rec.heterozygous_reference_minor_alleles_;
  } else {
    auto homozygous_offset = homozygousFilter(offset);   // computed and unused, as in the reference (:92)
    (void)homozygous_offset;
    if (!offset.getVariantArray().empty()) {
      auto unique_homozygous = uniqueUnphasedFilter(offset);
      rec.homozygous_minor_alleles_ += unique_homozygous->getVariantArray().size();
    }
    auto heterozygous_offset = heterozygousFilter(offset);
    rec.heterozygous_minor_alleles_ += heterozygous_offset->getVariantArray().size();
  }
}

std::map<std::string, std::map<std::string, VariantAnalysisType>> analyzeVariantPopulation(const PopulationDB& population) {
  std::map<std::string, std::map<std::string, VariantAnalysisType>> result;
  for (const auto& [genome_id, genome_ptr] : population.getMap()) {
    auto& contig_map = result[genome_id];
    for (const auto& [contig_id, contig_ptr] : genome_ptr->getMap()) {
      auto& contig_count = contig_map[contig_id];
      if (contig_ptr->variantCount() == 0) continue;
      for (const auto& [offset, offset_ptr] : contig_ptr->getMap()) updateVariantAnalysisType(*offset_ptr, contig_count);
    }
  }
  return result;
}

double wrightsFIS(const VariantAnalysisType& location, const VariantAnalysisType& genome) {
  double wrights_inbreeding = 0.0;
  if (location.total_variants_ > 0 && genome.total_variants_ > 0) {
    const double expected_heterozygosity =
        static_cast<double>(location.heterozygous_minor_alleles_ + location.heterozygous_reference_minor_alleles_) /
        static_cast<double>(location.total_variants_);
    const double observed_heterozygosity =
        static_cast<double>(genome.heterozygous_minor_alleles_ + genome.heterozygous_reference_minor_alleles_) /
        static_cast<double>(genome.total_variants_);
    wrights_inbreeding = (expected_heterozygosity - observed_heterozygosity) / expected_heterozygosity;
  }
  return wrights_inbreeding;
}

}  // namespace kgo
