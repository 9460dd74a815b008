"""-m gpu: the bench workload at BASELINE.json's full size (configs[2]: 10 M x 100 bp reads on a
chr20-shaped reference).  The oracle cannot run 10 M reads in a test, so the full-size run is checked
through size-independent properties of the path plus a sampled bit-exact comparison (tests/properties.py):

  * idempotence      -- searching the same device batch twice gives identical results
  * shard additivity -- reads are independent: the results of ragged contiguous shards, concatenated,
                        equal the result of the whole batch (what the multi-GPU path relies on)
  * structure        -- run-length-encoded lists are well formed for every read (lengths inside the
                        read, increasing along a list, far end only behind a close end, runs inside the
                        chromosome)
  * sampled parity   -- 20 000 reads drawn from all over the batch, bit-exact against the CPU oracle
"""
import pytest

from pindel_amd import synth
from tests.properties import check_workload

pytestmark = pytest.mark.gpu

CHR20_LEN = 62_435_964
N_READS = 10_000_000
READ_LEN = 100


def test_bench_workload_properties(engine_factory):
    import torch
    dev = torch.device("cuda", 0)
    ref = synth.make_reference(CHR20_LEN, seed=20260927, device=dev)
    chroms = [("20", ref)]
    batch = synth.make_reads(ref, N_READS, read_len=READ_LEN, seed=20260928, device=dev)
    eng = engine_factory()
    eng.load_reference(chroms)
    n_close, n_far = check_workload(eng, chroms, batch)
    assert n_close > 0.7 * N_READS and n_far > 0.5 * N_READS
    # The workload is seeded: the checksum of checksums bench.py prints as config.result_sha256 (one 64-bit digest per
    # read over its rc flag and run lists, sha256 over the digests) is a constant of the repository.  It has been
    # ec6c3a18... since the round-1 kernel; a kernel change that alters ANY read's result alters it.
    from pindel_amd import shard
    db = eng.upload(batch)
    eng.search_device(db)
    res = eng.download(db)
    eng.free_device_batch(db)
    assert int(res.close_off[-1]) + int(res.far_off[-1]) == 21235377
    assert int((res.far_off[1:] > res.far_off[:-1]).sum()) == 6290009
    assert shard.digest_hex(shard.read_digests(res)) == "ec6c3a18d3adc53802a769355a3129f4668632c3072f87679b6c1b245b94c19f"
