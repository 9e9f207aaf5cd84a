import sys, time
sys.path.insert(0,'/root/repo')
from tests import host_api as ha, records_io as rio, synth_vcf as sv, vcf_text as vt
G,L=2504,20000
rec,gt=sv.multiallelic_block(G,L,rng_seed=1,dup_records=0)
ids=[f"NA{i:05d}" for i in range(G)]
ref_text,dip_text=vt.write_vcf_mono(rec,"Gnomad2_1"),vt.write_vcf_1000(rec,gt,ids,quirks=False)
t0=time.perf_counter(); x=ha.InbreedInputs(ref_text,rio.DATA_SOURCE['Gnomad2_1'],dip_text); print("total",time.perf_counter()-t0)
