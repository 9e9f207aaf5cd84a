"""The sharded paths of the C ABI on the one GPU of the test box.  kgx_init takes a list of device slots; listing the
device twice (or three times) makes every handle split its genomes into that many shards with separate memory and
streams, swept by separate host threads, and the per-variant counts summed by the "peer" exchange.  What cannot run
here is RCCL across several devices; its call path (dlopen, ncclCommInitAll, grouped ncclAllReduce(sum, uint32) on the
library's streams) is exercised over ONE rank with KGX_EXCHANGE=rccl.  Everything is compared with the unsharded
run of the same population, which the other test files pin to the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture
def rebind(kgx, monkeypatch):
    """Bind other device lists inside a test; the session's one-device binding is restored afterwards."""
    def bind(devices, exchange=None):
        if exchange is None:
            monkeypatch.delenv("KGX_EXCHANGE", raising=False)
        else:
            monkeypatch.setenv("KGX_EXCHANGE", exchange)
        kgx.init(devices)
    yield bind
    monkeypatch.delenv("KGX_EXCHANGE", raising=False)
    kgx.init(0)
    assert kgx.bound_devices() == 1 and kgx.exchange_kind() == "none"


def dosage_results(kgx, G, V, codes, bins, n_bins, groups):
    pop = kgx.Population(G, V)
    pop.load_dosage2(kgx.pack_dosage2(codes))
    out = {
        "shards": pop.shards,
        "rows": pop.read_dosage2(),
        "k2": pop.allele_count_by_locus(),
        "k4": pop.population_summary(),
        "k3": pop.count_by_genome(),
        "k3_binned": pop.count_by_genome_binned(bins, n_bins),
        "k8": pop.compound_offsets(*groups, 3),
    }
    # the same groups as row lists; the FWS bins decided on the device from an AF column; then the population grown by
    # a third (rows uploaded afterwards) and swept again
    first, count, gbin = groups
    members = np.concatenate([np.arange(f, f + n, dtype=np.uint32) for f, n in zip(first, count)])
    starts = np.concatenate([[0], np.cumsum(count)[:-1]]).astype(np.uint32)
    out["k8_listed"] = pop.compound_offsets_listed(members, starts, count, gbin, 3)
    single_bin = (np.arange(V) % 3).astype(np.uint8)
    single_bin[members] = 0xFF
    out["offset_filters"] = pop.offset_filter_counts(single_bin, members, starts, count, gbin, 3)
    g_lo, g_hi = G // 5, G - G // 7
    out["row_lists_begin"], out["row_lists"] = pop.genome_row_lists(g_lo, g_hi, np.arange(V) % 3 != 1)
    selected = codes[(np.arange(V) % 3 != 1)][:, g_lo:g_hi] > 0
    assert int(out["row_lists_begin"][-1]) == int(selected.sum()) == len(out["row_lists"])
    af = (np.arange(V) % 23).astype(np.float32) / 40.0
    af[::17] = np.nan
    pop.set_af(af)
    out["k3_af_bins"] = pop.count_by_genome_af_bins([0.0, 0.05, 0.10, 0.15, 0.20, 0.25, 0.30, 0.35, 0.40, 0.45, 0.5, 1.0])
    # a genome mask (two genomes in three kept): a row counts as present when ANY shard holds a kept carrier of it
    keep = (np.arange(G) % 3 != 0).astype(np.uint8)
    pop.set_genome_mask(keep)
    out["k2_masked"] = pop.allele_count_by_locus()
    out["k3_masked"] = pop.count_by_genome()
    out["k3_binned_masked"] = pop.count_by_genome_binned(bins, n_bins)
    out["k3_af_bins_masked"] = pop.count_by_genome_af_bins([0.0, 0.05, 0.10, 0.15, 0.20, 0.25, 0.30, 0.35, 0.40, 0.45, 0.5, 1.0])
    out["k8_masked"] = pop.compound_offsets(*groups, 3)
    kept_codes = codes[:, keep.astype(bool)]
    assert np.array_equal(out["k2_masked"], np.stack([(kept_codes == k).sum(1) for k in range(4)], 1).astype(np.uint32))
    present = (kept_codes > 0).any(1) if kept_codes.shape[1] else np.zeros(V, dtype=bool)
    want_k3 = np.stack([(codes[present] == k).sum(0) for k in range(4)], 1).astype(np.uint64)
    want_k3[keep == 0] = 0
    assert np.array_equal(out["k3_masked"], want_k3)
    # the population summary leaves each shard's LOCAL counts in the buffer the by-genome sweep reads the population's
    # counts from (which rows still have a kept carrier): the sweep after it must count afresh, not trust that buffer
    out["offset_filters_masked"] = pop.offset_filter_counts(single_bin, members, starts, count, gbin, 3)
    assert np.array_equal(pop.allele_count_by_locus(), out["k2_masked"])
    out["k4_masked"] = pop.population_summary()
    assert np.array_equal(out["k4_masked"], out["k2_masked"].sum(0).astype(np.uint64))
    assert np.array_equal(pop.count_by_genome(), want_k3)
    assert np.array_equal(pop.offset_filter_counts(single_bin, members, starts, count, gbin, 3), out["offset_filters_masked"])
    pop.set_genome_mask(None)
    assert np.array_equal(pop.allele_count_by_locus(), out["k2"])
    extra = V // 3
    pop.resize(V + extra)
    pop.load_dosage2(kgx.pack_dosage2(codes[:extra]), V)
    out["k2_grown"] = pop.allele_count_by_locus()
    out["k3_grown"] = pop.count_by_genome()
    pop.resize(V)                                            # shrink, then grow within the allocation: the rows come back empty
    pop.resize(V + extra)
    regrown = pop.allele_count_by_locus()
    assert np.array_equal(regrown[:V], out["k2"]) and np.all(regrown[V:, 0] == G) and not regrown[V:, 1:].any()
    pop.close()
    return out


@pytest.mark.parametrize("slots", [2, 3])
@pytest.mark.parametrize("G", [1, 70, 1000, 4099])
def test_sharded_dosage_sweeps_equal_the_unsharded_ones(kgx, rebind, slots, G):
    rng = np.random.default_rng(G + slots)
    V = 1500
    codes = rng.choice(4, size=(V, G), p=[0.6, 0.25, 0.13, 0.02]).astype(np.uint8)
    bins = rng.integers(0, 11, V).astype(np.uint8)
    bins[rng.random(V) < 0.1] = 0xFF
    first = np.arange(0, V - 3, 7, dtype=np.uint32)
    groups = (first, np.full(len(first), 3, dtype=np.uint32), (first % 3).astype(np.uint32))
    want = dosage_results(kgx, G, V, codes, bins, 11, groups)
    assert len(want["shards"]) == 1
    rebind([0] * slots)
    assert kgx.bound_devices() == slots and kgx.exchange_kind() == "peer"
    got = dosage_results(kgx, G, V, codes, bins, 11, groups)
    shards = got["shards"]
    assert len(shards) == slots and sum(s["n_genomes"] for s in shards) == G
    assert all(s["genome_base"] % 64 == 0 for s in shards if s["n_genomes"])
    assert [s["genome_base"] for s in shards] == list(np.cumsum([0] + [s["n_genomes"] for s in shards[:-1]]))
    for key in ("rows", "k2", "k4", "k3", "k3_binned", "k8", "k8_listed", "k3_af_bins", "k2_grown", "k3_grown", "k2_masked", "k3_masked",
                "k3_binned_masked", "k3_af_bins_masked", "k8_masked", "k4_masked", "offset_filters_masked", "offset_filters", "row_lists_begin", "row_lists"):
        assert np.array_equal(got[key], want[key]), key
    assert np.array_equal(want["k8_listed"], want["k8"])
    assert np.array_equal(want["k2_grown"][:V], want["k2"]) and np.array_equal(want["k2_grown"][V:], want["k2"][:V // 3])


@pytest.mark.parametrize("binding", [2, 4, 8, "all"])
def test_distinct_devices_exchange_over_rccl(kgx, rebind, binding):
    """kgx_init over n DISTINCT devices -- the first 2, 4, 8 ordinals, and kgx_init(0, NULL): every visible one --: communicators
    from ncclCommInitAll over all of them, the K2 counts summed by a grouped in-place ncclAllReduce from one thread on the slots'
    streams (Exchange::Rccl with n > 1) -- what a multi-GPU node runs.  K2/K3/K4/K8, the masked sweeps, the row lists, the
    inbreeding sweeps (window-sized, batched, and on the moments) against the unsharded results.  A case is skipped where the box
    has fewer devices (a one-GPU box skips all four: the multi-rank RCCL path is then NOT verified on hardware)."""
    visible = kgx.device_count()
    n = visible if binding == "all" else binding
    if visible < 2 or n > visible:
        pytest.skip(f"needs {max(n, 2)} visible MI355X devices, the box shows {visible}")
    rng = np.random.default_rng(11)
    G, V = 4099, 1500
    codes = rng.choice(4, size=(V, G), p=[0.6, 0.25, 0.13, 0.02]).astype(np.uint8)
    bins = rng.integers(0, 11, V).astype(np.uint8)
    first = np.arange(0, V - 3, 7, dtype=np.uint32)
    groups = (first, np.full(len(first), 3, dtype=np.uint32), (first % 3).astype(np.uint32))
    want = dosage_results(kgx, G, V, codes, bins, 11, groups)
    Gi, Li = 2100, 12_000                                   # (more than 8192 loci: HallME and Loglikelihood on per-genome moments)
    one = kgx.GenotypeMatrix(Gi, Li)
    table = one.synth_multiallelic(1111, 0, 0)
    window = np.arange(0, Li, 12, dtype=np.uint32)
    tasks = [{"g0": 0, "g1": 900, "locus_index": window + k, "minor_af": np.ascontiguousarray(table[window + k])} for k in range(3)] + \
            [{"g0": 896, "g1": Gi, "locus_index": window, "minor_af": np.ascontiguousarray(table[window])}]
    want_inbreed = {a: one.inbreed(table, a, phased=True) for a in ("Simple", "HallME", "Loglikelihood")}
    want_batch = {a: one.inbreed_batch(tasks, a, phased=True) for a in ("RitlandLocus", "Loglikelihood")}
    one.close()
    rebind([] if binding == "all" else list(range(n)))
    assert kgx.bound_devices() == n and kgx.exchange_kind() == "rccl"
    got = dosage_results(kgx, G, V, codes, bins, 11, groups)
    assert len(got["shards"]) == n and {s["slot"] for s in got["shards"]} == set(range(n))
    for key in want:
        if key != "shards":
            assert np.array_equal(got[key], want[key]), key
    many = kgx.GenotypeMatrix(Gi, Li)
    many.synth_multiallelic(1111, 0, 0)
    assert len(many.shards) == n

    def same(result, wanted, ctx):
        for name in wanted.dtype.names:
            if name.endswith("_count"):
                assert np.array_equal(result[name], wanted[name]), ctx + (name,)
            else:
                assert np.allclose(result[name], wanted[name], rtol=1e-11, atol=1e-11), ctx + (name,)

    for algorithm, wanted in want_inbreed.items():
        same(many.inbreed(table, algorithm, phased=True), wanted, (algorithm,))
    for algorithm, wanted in want_batch.items():
        for k, (result, w) in enumerate(zip(many.inbreed_batch(tasks, algorithm, phased=True), wanted)):
            same(result, w, (algorithm, "batch", k))
    many.close()


def test_sharded_loaders_and_synthetic_population(kgx, rebind):
    G, V = 3000, 4000
    one = kgx.Population(G, V)
    one.synth_biallelic(1111, 500, 10)
    want_rows, want_af, want_k2 = one.read_dosage2(), one.get_af(), one.allele_count_by_locus()
    dosage = np.ascontiguousarray(kgx.unpack_dosage2(want_rows, G).T)                  # the reference's [G][V] uint8 rows
    one.close()
    rebind([0, 0, 0])
    many = kgx.Population(G, V)
    many.synth_biallelic(1111, 500, 10)                          # every shard draws its own genomes of the same population
    assert np.array_equal(many.read_dosage2(), want_rows) and np.array_equal(many.get_af(), want_af)
    assert np.array_equal(many.allele_count_by_locus(), want_k2)
    many.close()
    loaded = kgx.Population(G, V)
    loaded.load_dosage_u8(dosage[:1028], 0)                      # genome-major uploads that straddle shard boundaries
    loaded.load_dosage_u8(dosage[1028:], 1028)
    assert np.array_equal(loaded.read_dosage2(), want_rows)
    assert np.array_equal(loaded.allele_count_by_locus(), want_k2)
    loaded.close()


def test_counts_left_on_the_device_are_exchanged_too(kgx, rebind):
    """kgx_allele_count_by_locus_dev / kgx_allele_frequency_dev (what bench.py drives) on a sharded population: the
    caller's buffer on the first slot's device receives the summed counts on the caller's stream."""
    import torch

    G, V = 5000, 20_000
    one = kgx.Population(G, V)
    one.synth_biallelic(1111, 0, 0)
    want = one.allele_count_by_locus()
    one.close()
    rebind([0, 0])
    pop = kgx.Population(G, V)
    pop.synth_biallelic(1111, 0, 0)
    dev = torch.device("cuda", 0)
    counts = torch.zeros((V, 4), dtype=torch.int32, device=dev)
    af = torch.zeros(V, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(3):
        pop.allele_count_by_locus_dev(counts.data_ptr(), stream)
        kgx.allele_frequency_dev(counts.data_ptr(), V, G, af.data_ptr(), stream)
    torch.cuda.synchronize(dev)
    got = counts.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, want)
    assert np.array_equal(af.cpu().numpy(), (want[:, 1].astype(np.float64) + 2.0 * want[:, 2]) / (2.0 * G))
    ms = pop.allele_count_timed(counts.data_ptr(), stream, 1, 3)
    assert ms.shape == (3,) and np.all(ms > 0)
    pop.close()


def test_rccl_all_reduce_call_path_over_one_rank(kgx, rebind):
    """KGX_EXCHANGE=rccl on one device: librccl is loaded, a communicator created (ncclCommInitAll) and every count sweep
    goes through ncclAllReduce(sum, uint32) on the library's stream -- over a single rank, so the sums are the shard's own."""
    G, V = 2500, 30_000
    plain = kgx.Population(G, V)
    plain.synth_biallelic(1111, 0, 0)
    want = plain.allele_count_by_locus()
    plain.close()
    rebind([0], exchange="rccl")
    assert kgx.exchange_kind() == "rccl" and kgx.bound_devices() == 1
    pop = kgx.Population(G, V)
    pop.synth_biallelic(1111, 0, 0)
    for _ in range(3):
        assert np.array_equal(pop.allele_count_by_locus(), want)
    pop.close()


@pytest.mark.parametrize("algorithm", ["Simple", "RitlandLocus", "HallME", "Loglikelihood"])
def test_sharded_inbreeding_equals_the_unsharded_one(kgx, rebind, algorithm, monkeypatch):
    G, L = 700, 3000
    one = kgx.GenotypeMatrix(G, L)
    table = one.synth_multiallelic(1111, 0, 0)
    want_rows = one.read_rows()
    index = np.arange(0, L, 2, dtype=np.uint32)
    sub = np.ascontiguousarray(table[index])
    want_all = one.inbreed(table, algorithm, phased=True)
    want_part = one.inbreed(sub, algorithm, phased=True, locus_index=index, g0=100, g1=650)
    # ... the same through the multi-kernel paths (the iterative estimators on per-genome moments), and as a batch of tasks
    monkeypatch.setenv("KGX_K7_NO_WAVE", "1")
    want_big = one.inbreed(table, algorithm, phased=True)
    big_path = kgx.inbreed_last_path()
    monkeypatch.delenv("KGX_K7_NO_WAVE")
    tasks = [{"g0": 0, "g1": 300, "locus_index": index, "minor_af": sub}, {"g0": 296, "g1": G, "locus_index": index[::2], "minor_af": sub[::2]},
             {"g0": 100, "g1": 650, "locus_index": None, "minor_af": np.ascontiguousarray(table[:1000])}]
    want_batch = one.inbreed_batch(tasks, algorithm, phased=True)
    one.close()
    rebind([0, 0, 0])
    many = kgx.GenotypeMatrix(G, L)
    shards = many.shards
    assert len(shards) == 3 and sum(s["n_genomes"] for s in shards) == G and all(s["genome_base"] % 128 == 0 for s in shards if s["n_genomes"])
    table2 = many.synth_multiallelic(1111, 0, 0)
    assert np.array_equal(many.read_rows(), want_rows)
    assert np.array_equal(np.nan_to_num(table2), np.nan_to_num(table))
    got_all = many.inbreed(table, algorithm, phased=True)
    got_part = many.inbreed(sub, algorithm, phased=True, locus_index=index, g0=100, g1=650)      # a range that straddles shards
    monkeypatch.setenv("KGX_K7_NO_WAVE", "1")
    got_big = many.inbreed(table, algorithm, phased=True)
    assert kgx.inbreed_last_path() == big_path and big_path == {"HallME": "hall moments", "Loglikelihood": "loglik moments"}.get(algorithm, "frequency sweep")
    monkeypatch.delenv("KGX_K7_NO_WAVE")
    got_batch = many.inbreed_batch(tasks, algorithm, phased=True)
    for got, want in [(got_all, want_all), (got_part, want_part), (got_big, want_big)] + list(zip(got_batch, want_batch)):
        for name in want.dtype.names:
            if name.endswith("_count"):
                assert np.array_equal(got[name], want[name]), name
            else:   # a genome's fp64 sums do not depend on which shard holds it: its lane's arithmetic is the same
                assert np.allclose(got[name], want[name], rtol=1e-11, atol=1e-11), name
    # host uploads split across the shards as well
    loaded = kgx.GenotypeMatrix(G, L)
    loaded.load_rows(want_rows)
    assert np.array_equal(loaded.read_rows(), want_rows)
    loaded.load_genomes(np.ascontiguousarray(want_rows.T[:300]), 0)
    loaded.load_genomes(np.ascontiguousarray(want_rows.T[300:]), 300)
    assert np.array_equal(loaded.read_rows(), want_rows)
    many.close(); loaded.close()


def test_handles_outlive_a_rebinding(kgx, rebind):
    """A handle keeps the binding it was created under: created sharded, used after the library was rebound."""
    rebind([0, 0])
    pop = kgx.Population(900, 1000)
    pop.synth_biallelic(1111, 0, 0)
    kgx.init(0)
    single = kgx.Population(900, 1000)
    single.synth_biallelic(1111, 0, 0)
    assert len(pop.shards) == 2 and len(single.shards) == 1
    assert np.array_equal(pop.allele_count_by_locus(), single.allele_count_by_locus())
    assert np.array_equal(pop.count_by_genome(), single.count_by_genome())
    pop.close(); single.close()


def test_bench_exchange_calls_run_on_rccl_with_one_rank(tmp_path):
    """What bench.py does between ranks -- torch.distributed on the nccl (= RCCL) backend bound to the device, an
    asynchronous all_reduce(SUM) of the int32 view of the uint32 counts, work.wait() on the current stream, barrier --
    run with a one-rank group in a child process (a box with one GPU cannot hold two RCCL ranks): the calls, dtypes and
    stream semantics are RCCL's, only the peers are missing."""
    import subprocess
    import sys
    import textwrap

    script = tmp_path / "one_rank.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        import numpy as np
        import torch
        import torch.distributed as dist
        sys.path.insert(0, os.environ["KGX_ROOT"])
        from kgl_gene_amd import capi
        from kgl_gene_amd.sharding import allreduce_counts_async
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", TORCH_NCCL_HIGH_PRIORITY="1")
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", device_id=dev)
        capi.ensure_built(); capi.init(0)
        G, V = 4000, 50_000
        pop = capi.Population(G, V); pop.synth_biallelic(1111, 0, 0)
        want = pop.allele_count_by_locus()
        bufs = [torch.empty((V, 4), dtype=torch.int32, device=dev) for _ in range(2)]
        af = torch.empty((V,), dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        pending = None
        for i in range(4):                                   # bench.py's pipeline: sweep i+1 beside the exchange of i
            buf = bufs[i % 2]
            pop.allele_count_by_locus_dev(buf.data_ptr(), stream)
            work = allreduce_counts_async(buf, 2)            # world size "2": issue the collective (this group has one rank)
            if pending is not None:
                pending[0].wait()
                capi.allele_frequency_dev(pending[1].data_ptr(), V, G, af.data_ptr(), stream)
            pending = (work, buf)
        pending[0].wait()
        capi.allele_frequency_dev(pending[1].data_ptr(), V, G, af.data_ptr(), stream)
        dist.barrier()
        torch.cuda.synchronize(dev)
        got = pending[1].cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want)
        assert np.array_equal(af.cpu().numpy(), (want[:, 1].astype(np.float64) + 2.0 * want[:, 2]) / (2.0 * G))
        dist.destroy_process_group()
        print("one-rank RCCL exchange ok")
    """))
    import os
    from pathlib import Path

    env = dict(os.environ, KGX_ROOT=str(Path(__file__).resolve().parent.parent), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0 and "one-rank RCCL exchange ok" in res.stdout, res.stderr[-3000:]


@pytest.mark.parametrize("workload", ["c3", "c5"])
def test_bench_two_ranks_rehearsal_on_one_device(tmp_path, workload):
    """bench.py's N > 1 path end to end, as the driver launches it (torch.distributed.run, one process per rank), on the
    one GPU of the test box: both ranks sweep their genome shard with the HIP kernels on device 0 and the counts are
    exchanged through gloo (KGX_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device -- its calls are covered by the
    one-rank test above).  The line must be rank 0's only, carry both ranks' work and pass its own exchange check."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, KGX_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(key, None)
    shape = ["--genomes", "3000", "--variants", "300000", "--c4-genomes", "5001"] if workload == "c3" else ["--workload", "c5", "--genomes", "1500", "--variants", "100000"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           "29541" if workload == "c3" else "29542", str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", *shape]
    res = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [line for line in res.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, res.stdout
    record = json.loads(lines[0])
    assert record["n_gpus"] == 2 and record["steps"] == 3 and record["warmup"] == 1 and record["value"] > 0
    assert record["scaling"] == "weak" and record["vs_baseline"] is None
    if workload == "c3":
        assert record["config"]["total_genomes"] == 2 * record["config"]["genomes_per_gpu"] == 6000
        assert record["config"]["exchange"].startswith("gloo rehearsal") and "MISMATCH" not in record["config"]["exchange_check"]
        assert "all 300000 variants" in record["config"]["exchange_check"]
        distributed = dict(record["config"]["distributed"])
        spread = distributed.pop("over_ranks")         # every rank's own clocks: the kernel, the step, what the exchange leaves exposed
        assert distributed == {"world_size": 2, "backend": "gloo", "kgx_exchange_kind": "none", "kgx_bound_devices": 1}
        assert 0 < spread["kernel_ms_min"] <= spread["kernel_ms_max"] <= spread["step_ms_max"] and spread["step_ms_min"] <= spread["step_ms_max"]
        assert abs(spread["exposed_exchange_and_epilogue_ms"] - (spread["step_ms_max"] - spread["kernel_ms_max"])) <= 1e-6 * spread["step_ms_max"] + 1e-9
        # ... and north_star's strong-scaled job beside it: one population split over the ranks, the same step
        strong = record["aux"]["c4_strong"]
        assert strong["scaling"] == "strong" and strong["n_gpus"] == 2 and strong["config"]["total_genomes"] == 5001
        assert strong["config"]["genomes_per_gpu"] in (2500, 2501, 2504) and "MISMATCH" not in strong["config"]["exchange_check"]
        assert abs(strong["value"] - 5001 * 300000 * 3 / (strong["ms_per_step"] * 3e-3)) <= 1e-6 * strong["value"]
        assert abs(record["value"] - 6000 * 300000 * 3 / (record["ms_per_step"] * 3e-3)) <= 1e-6 * record["value"]
