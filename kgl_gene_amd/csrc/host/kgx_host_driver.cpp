// Stand-alone driver for the GPU analysis packages: plays the role of ExecutePackage
// (kgl_app/kgl_package.cpp:17-79) for ONE package — builds the data objects a parser would deliver from a
// small binary record file, then calls initializeAnalysis -> fileReadAnalysis (per file) -> iterationAnalysis
// -> finalizeAnalysis on the VirtualAnalysis interface.  Used by the parity tests; in the reference tree the
// packages are registered in the factory map instead.
//
//   kgx_host_driver <IDENT> <work_dir> [key=value ...] -- <records.bin | vcf:[<DataSource>:]<file.vcf> | ped:<file>
//                                                            | pf7sample:<file.tsv> | pf7fws:<file.tsv>> ...
//
// "vcf:<path>" hands the package a FilenameDataDB (the reference's "FileNameOnly" data file,
// kgl_parser/kgl_variant_factory_parsers.cpp:65-66): the package reads the VCF itself.  "pf7sample:" / "pf7fws:" load
// the Pf7 sample and FWS resources (kgl_app/kgl_package_resource_pf.cpp) the way the runtime XML's resource entries do.
//
// Record file (little endian): "KGXR" u32 version=1, u32 mode (0 = phased 1000-Genomes style, 1 = unphased
// Pf style, 2 = reference mono-genome), u32 data_source (DataSourceEnum), str population_id, str contig,
// u64 n_genomes, n_genomes x str id, u64 n_records, then per record: u64 offset, str ref, u8 n_alt, n_alt x str,
// u8 pass, u8 n_info_fields, per field: str name, u32 n, n x f32; finally the genotype matrix
// u8[n_records][n_genomes][2] (allele indices, 0 = reference; absent for mode 2); then optionally
// u64 n_ped, n_ped x (str genome, str super_population).      str = u32 length + bytes.
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>

#include "kga_analysis_gpu_allele.h"
#if __has_include("kga_analysis_gpu_inbreed.h")
#include "kga_analysis_gpu_inbreed.h"
#define KGX_HAVE_INBREED_PACKAGE 1
#endif

namespace kgl = kellerberrin::genome;
namespace kga = kellerberrin::genome::analysis;
using kellerberrin::ExecEnv;

namespace {

struct Reader {
  std::ifstream in;
  template <typename T> T pod() { T v{}; in.read(reinterpret_cast<char*>(&v), sizeof(T)); return v; }
  std::string str() { const uint32_t n = pod<uint32_t>(); std::string s(n, '\0'); in.read(s.data(), n); return s; }
};

struct LoadedFile {
  std::shared_ptr<kgl::DataDB> population;
  std::vector<std::pair<std::string, std::string>> ped;
};

LoadedFile loadRecords(const std::string& path) {
  Reader r;
  r.in.open(path, std::ios::binary);
  if (!r.in.good()) ExecEnv::log().critical("cannot open record file: {}", path);
  char magic[4];
  r.in.read(magic, 4);
  if (std::memcmp(magic, "KGXR", 4) != 0 || r.pod<uint32_t>() != 1) ExecEnv::log().critical("bad record file: {}", path);
  const uint32_t mode = r.pod<uint32_t>();
  const auto data_source = static_cast<kgl::DataSourceEnum>(r.pod<uint32_t>());
  const std::string population_id = r.str();
  const std::string contig = r.str();
  const uint64_t G = r.pod<uint64_t>();
  std::vector<std::string> ids(G);
  for (auto& id : ids) id = r.str();
  const uint64_t R = r.pod<uint64_t>();

  struct Rec { uint64_t offset; std::string ref; std::vector<std::string> alts; bool pass; std::shared_ptr<kgl::InfoRecord> info; };
  std::vector<Rec> recs(R);
  for (auto& rec : recs) {
    rec.offset = r.pod<uint64_t>();
    rec.ref = r.str();
    rec.alts.resize(r.pod<uint8_t>());
    for (auto& a : rec.alts) a = r.str();
    rec.pass = r.pod<uint8_t>() != 0;
    rec.info = std::make_shared<kgl::InfoRecord>();
    const uint8_t n_fields = r.pod<uint8_t>();
    for (uint8_t f = 0; f < n_fields; ++f) {
      const std::string name = r.str();
      std::vector<float> v(r.pod<uint32_t>());
      for (auto& x : v) x = r.pod<float>();
      rec.info->float_fields.emplace(name, std::move(v));
    }
  }
  std::vector<uint8_t> gt;
  if (mode != 2) {
    gt.resize(R * G * 2);
    r.in.read(reinterpret_cast<char*>(gt.data()), static_cast<std::streamsize>(gt.size()));
  }
  LoadedFile out;
  if (r.in.good() && r.in.peek() != EOF) {
    const uint64_t n_ped = r.pod<uint64_t>();
    for (uint64_t i = 0; i < n_ped; ++i) { auto g = r.str(); auto sp = r.str(); out.ped.emplace_back(g, sp); }
  }

  auto pop = std::make_shared<kgl::PopulationDB>(population_id, data_source);
  if (mode == 1 || mode == 2) for (const auto& id : ids) (void)pop->getCreateGenome(id);   // setupPopulationStructure (pf_impl.cpp:399-425)
  auto make = [&](const Rec& rec, size_t record_index, uint32_t alt, kgl::VariantPhase phase) {
    kgl::VariantEvidence evidence(record_index, data_source, rec.pass, rec.info, alt, static_cast<uint32_t>(rec.alts.size()));
    return std::make_shared<const kgl::Variant>(contig, rec.offset, phase, "", kgl::DNA5SequenceLinear(rec.ref),
                                                kgl::DNA5SequenceLinear(rec.alts[alt]), evidence);
  };
  for (size_t ri = 0; ri < R; ++ri) {
    const Rec& rec = recs[ri];
    const uint32_t A = static_cast<uint32_t>(rec.alts.size());
    if (mode == 2) {
      for (uint32_t a = 0; a < A; ++a) pop->addVariant(make(rec, ri, a, kgl::VariantPhase::UNPHASED), {ids[0]});
    } else if (mode == 0) {   // Genome1000VCFImpl::ParseRecord (1000_impl.cpp:63-145): shared Variant per (alt, phase), A then B
      for (int phase = 0; phase < 2; ++phase) {
        std::map<size_t, std::vector<kgl::GenomeId_t>> phase_map;
        for (uint64_t g = 0; g < G; ++g) {
          const uint32_t idx = gt[(ri * G + g) * 2 + phase];
          if (idx != 0 && idx <= A) phase_map[idx - 1].push_back(ids[g]);
        }
        for (const auto& [alt, genomes] : phase_map)
          pop->addVariant(make(rec, ri, static_cast<uint32_t>(alt), phase == 0 ? kgl::VariantPhase::DIPLOID_PHASE_A : kgl::VariantPhase::DIPLOID_PHASE_B), genomes);
      }
    } else {                  // PfVCFImpl (pf_impl.cpp:287-384): a fresh UNPHASED Variant per allele copy
      for (uint64_t g = 0; g < G; ++g)
        for (int copy = 0; copy < 2; ++copy) {
          const uint32_t idx = gt[(ri * G + g) * 2 + copy];
          if (idx != 0 && idx <= A) pop->addVariant(make(rec, ri, idx - 1, kgl::VariantPhase::UNPHASED), {ids[g]});
        }
    }
  }
  out.population = pop;
  return out;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 5) {
    std::cerr << "usage: kgx_host_driver <IDENT> <work_dir> [key=value ...] -- <records.bin> ...\n";
    return 2;
  }
  const std::string ident = argv[1];
  const std::string work_dir = argv[2];
  kgl::ParameterMap parameters;
  int i = 3;
  for (; i < argc && std::strcmp(argv[i], "--") != 0; ++i) {
    const std::string kv = argv[i];
    const auto eq = kv.find('=');
    if (eq == std::string::npos) { std::cerr << "bad parameter: " << kv << "\n"; return 2; }
    if (kv.substr(0, eq) == "quiet") ExecEnv::log().quiet(true);
    else parameters.insert(kv.substr(0, eq), kv.substr(eq + 1));
  }
  ++i;

  // The factory map a maintainer extends in kga_analytic/kga_analysis_factory.cpp:31-43.
  const kgl::VirtualAnalysis::AnalysisFactoryMap factory_map = {
      {kga::GpuAlleleAnalysis::IDENT, kga::GpuAlleleAnalysis::factory},
#ifdef KGX_HAVE_INBREED_PACKAGE
      {kga::GpuInbreedAnalysis::IDENT, kga::GpuInbreedAnalysis::factory},
#endif
  };
  auto factory = factory_map.find(ident);
  if (factory == factory_map.end()) { std::cerr << "unknown analysis ident: " << ident << "\n"; return 2; }
  std::unique_ptr<kgl::VirtualAnalysis> analysis = factory->second();

  std::vector<LoadedFile> files;
  auto genealogy = std::make_shared<kgl::HsGenomeGenealogyData>("PED");
  std::vector<std::shared_ptr<const kgl::ResourceBase>> pf7_resources;
  for (; i < argc; ++i) {
    if (std::strncmp(argv[i], "vcf:", 4) == 0) {
      // vcf:<path>  or  vcf:<DataSource>:<path>  (the data source the runtime XML gives the file)
      static const std::map<std::string, kgl::DataSourceEnum> sources{
          {"Genome1000", kgl::DataSourceEnum::Genome1000}, {"GnomadGenome3_1", kgl::DataSourceEnum::GnomadGenome3_1},
          {"Falciparum", kgl::DataSourceEnum::Falciparum}, {"GnomadExomes3_1", kgl::DataSourceEnum::GnomadExomes3_1},
          {"GnomadExomes2_1", kgl::DataSourceEnum::GnomadExomes2_1}, {"Gnomad3_1", kgl::DataSourceEnum::Gnomad3_1},
          {"Gnomad3_0", kgl::DataSourceEnum::Gnomad3_0}, {"Gnomad2_1", kgl::DataSourceEnum::Gnomad2_1}};
      std::string rest(argv[i] + 4);
      kgl::DataSourceEnum source = kgl::DataSourceEnum::NotImplemented;
      const size_t colon = rest.find(':');
      if (colon != std::string::npos) {
        auto it = sources.find(rest.substr(0, colon));
        if (it != sources.end()) { source = it->second; rest = rest.substr(colon + 1); }
      }
      LoadedFile f;
      f.population = std::make_shared<kgl::FilenameDataDB>(source, rest);
      files.push_back(std::move(f));
      continue;
    }
    if (std::strncmp(argv[i], "ped:", 4) == 0) {
      // The genealogy resource.  Either the reference's PED file -- a header line, then 15 tab-separated fields per sample
      // (ParseHsGenomeGenealogyFile::moveToRecord, kgl_parser/kgl_hsgenealogy_parser.cpp) -- or lines of "<genome>\t<super population>".
      std::ifstream in(argv[i] + 4);
      std::string line;
      bool first_line = true;
      while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        std::vector<std::string> fields;
        size_t begin = 0;
        for (size_t tab = line.find('\t'); ; tab = line.find('\t', begin)) {
          fields.push_back(line.substr(begin, tab == std::string::npos ? std::string::npos : tab - begin));
          if (tab == std::string::npos) break;
          begin = tab + 1;
        }
        if (fields.size() == kgl::HsGenealogyRecord::genealogyFieldCount()) {
          if (!first_line)      // the reference skips the header row
            genealogy->addGenealogyRecord(kgl::HsGenealogyRecord(fields[0], fields[1], fields[2], fields[3], fields[4], fields[5], fields[6], fields[7],
                                                                 fields[8], fields[9], fields[10], fields[11], fields[12], fields[13], fields[14]));
        } else if (fields.size() == 2) {
          genealogy->addGenealogyRecord(kgl::HsGenealogyRecord(fields[0], fields[1]));
        } else {
          ExecEnv::log().error("ParseHsGenomeGenealogyFile::moveToRecord; field count: {} not equal mandatory count: {}", fields.size(),
                               kgl::HsGenealogyRecord::genealogyFieldCount());
        }
        first_line = false;
      }
      continue;
    }
    if (std::strncmp(argv[i], "pf7sample:", 10) == 0) {
      kgl::ParsePf7Sample parser;
      if (!parser.parsePf7SampleFile(argv[i] + 10)) { std::cerr << "cannot read the Pf7 sample file " << (argv[i] + 10) << "\n"; return 2; }
      pf7_resources.push_back(std::make_shared<kgl::Pf7SampleResource>("Pf7SampleDriver", parser.getPf7SampleVector()));
      continue;
    }
    if (std::strncmp(argv[i], "pf7fws:", 7) == 0) {
      kgl::ParsePf7Fws parser;
      if (!parser.parsePf7FwsFile(argv[i] + 7)) { std::cerr << "cannot read the Pf7 FWS file " << (argv[i] + 7) << "\n"; return 2; }
      pf7_resources.push_back(std::make_shared<kgl::Pf7FwsResource>("Pf7FwsDriver", parser.getPf7FwsVector()));
      continue;
    }
    files.push_back(loadRecords(argv[i]));
    for (const auto& [genome, sp] : files.back().ped) genealogy->addGenealogyRecord(kgl::HsGenealogyRecord(genome, sp));
  }
  auto resources = std::make_shared<kgl::AnalysisResources>();
  resources->addResource(genealogy);
  for (const auto& resource : pf7_resources) resources->addResource(resource);

  kgl::ActiveParameterList named_parameters;
  named_parameters.addNamedParameterVector({"DriverParameters", kgl::ParameterVector{parameters}});

  // Sequencing and error convention of PackageAnalysis (kgl_app/kgl_package_analysis.cpp): false disables the analysis.
  if (!analysis->initializeAnalysis(work_dir, named_parameters, resources)) { std::cerr << "initializeAnalysis failed\n"; return 1; }
  for (const auto& f : files)
    if (!analysis->fileReadAnalysis(f.population)) { std::cerr << "fileReadAnalysis failed\n"; return 1; }
  if (!analysis->iterationAnalysis()) { std::cerr << "iterationAnalysis failed\n"; return 1; }
  if (!analysis->finalizeAnalysis()) { std::cerr << "finalizeAnalysis failed\n"; return 1; }
  return 0;
}
