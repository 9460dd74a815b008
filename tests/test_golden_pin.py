"""Pins the search against the reference's OWN golden files (devtools/gold_standard/simulated_test.*,
committed under tests/golden/sim1chrVs2): reads -> search -> C++ classifiers/reporters -> the four
reports must be byte-identical to the gold standard (coverage columns masked).

  * test_oracle_pinned_by_gold_reports (CPU): the search is the CPU oracle -> this is what pins the
    oracle that every other parity test trusts.
  * test_gpu_path_reproduces_gold_reports (-m gpu): the search is the HIP path through the C ABI.
"""
import numpy as np
import pytest

from oracle import pyoracle
from pindel_amd import hostio, hostlib
from tests import golden_util as gu


def _load(tmp_path):
    fa, reads_txt = gu.unpack(tmp_path)
    chroms = hostio.load_fasta(fa)
    batch = hostio.read_pindel_text(reads_txt, [n for n, _ in chroms], [len(s) - 200000 for _, s in chroms])
    assert batch.n == 14862 and len(chroms[0][1]) == 400001
    return fa, reads_txt, chroms, batch


def test_oracle_pinned_by_gold_reports(tmp_path):
    fa, reads_txt, chroms, batch = _load(tmp_path)
    p = pyoracle.make_params()
    r = pyoracle.search_batch(p, [s for _, s in chroms], batch.seq, batch.seq_off, batch.anchor_strand,
                              batch.anchor_pos, batch.insert_size, batch.chr_id)
    # figures the survey measured with the reference binary itself
    assert int((r["close_cnt"] > 0).sum()) == 14862
    assert int((r["far_cnt"] > 0).sum()) == 10968
    assert int(r["rc_flag"].sum()) == 30
    co, cp = gu.csr_from_strided(r["close_cnt"], r["close_pts"])
    fo, fp = gu.csr_from_strided(r["far_cnt"], r["far_pts"])
    st = hostlib.default_settings(pyoracle.max_mismatch_table())
    prefix = str(tmp_path / "oracle")
    hostlib.call_from_points(fa, reads_txt, prefix, st, co, cp, fo, fp, r["rc_flag"])
    gu.assert_reports_match_gold(prefix)


@pytest.mark.gpu
def test_gpu_path_reproduces_gold_reports(tmp_path, engine_factory):
    from pindel_amd import binding
    fa, reads_txt, chroms, batch = _load(tmp_path)
    eng = engine_factory()
    eng.load_fasta(fa)                      # the library's own FASTA loader
    assert eng.reference_info() == [("1", 400001)]
    res = eng.search_batch(batch)
    assert int((res.close_off[1:] > res.close_off[:-1]).sum()) == 14862
    assert int((res.far_off[1:] > res.far_off[:-1]).sum()) == 10968
    assert int(res.rc_flag.sum()) == 30
    cp = binding.expand_runs(res.close_runs)
    fp = binding.expand_runs(res.far_runs)
    # CSR over points from CSR over runs
    def point_off(off, runs):
        per_run = (runs["len_last"].astype(np.int64) - runs["len_first"].astype(np.int64) + 1)
        cum = np.concatenate([[0], np.cumsum(per_run)])
        return cum[off.astype(np.int64)].astype(np.uint64)
    co = point_off(res.close_off, res.close_runs)
    fo = point_off(res.far_off, res.far_runs)
    st = hostlib.default_settings(eng.max_mismatch_table())
    prefix = str(tmp_path / "gpu")
    hostlib.call_from_points(fa, reads_txt, prefix, st, co, cp, fo, fp, res.rc_flag)
    gu.assert_reports_match_gold(prefix)


@pytest.mark.gpu
def test_command_line_reproduces_gold_reports(tmp_path):
    """pindel_pg (C++ host + C ABI): -f/-p/-o like `pindel -f ... -p ... -o ...`."""
    import os
    import subprocess
    from pindel_amd import binding
    fa, reads_txt = gu.unpack(tmp_path)
    exe = os.path.join(os.path.dirname(binding.LIB_PATH), "pindel_pg")
    prefix = str(tmp_path / "cli")
    out = subprocess.run([exe, "-f", fa, "-p", reads_txt, "-o", prefix, "-T", "1"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "close end 14862, far end 10968" in out.stdout
    gu.assert_reports_match_gold(prefix)
