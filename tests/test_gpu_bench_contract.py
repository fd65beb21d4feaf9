"""bench.py's output contract (the round driver parses this line): one JSON line on stdout with the agreed keys, the
`roofline` and `cpu_baseline` objects, recall parity, on a reduced workload so that the test takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, with_detail=False):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "64",
                          "--corpus", "30000", *extra], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    all_lines = out.stdout.splitlines()
    lines = [l for l in all_lines if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    assert all_lines[-1] == lines[0], "the JSON line is the LAST line of stdout"
    # the round driver keeps the last 8 KB of stdout: the headline must fit with room to spare (round 4's line was 20 KB and was lost)
    assert len(lines[0]) <= 4096, len(lines[0])
    j = json.loads(lines[0])
    if not with_detail:
        return j
    det = [l for l in all_lines if l.startswith("#stages ")]
    assert len(det) == 1
    # both lines together fit the driver's 8 KB stdout tail (a tail that begins inside the #stages line could start with a `{`)
    assert len(det[0]) + len(lines[0]) + 2 <= 8000, (len(det[0]), len(lines[0]))
    short = json.loads(det[0][len("#stages "):])
    assert "stages" in short and "generate" in short["stages"] and "similarity_topk_f32" in short["stages"]
    return j, json.load(open(os.path.join(ROOT, "bench_stages.json")))          # the full object, with its prose


def test_bench_line_contract_fp32():
    """The DEFAULT stage list (exactly what `python bench.py` runs, on a reduced corpus): the line's shape and length are those of
    the full-size run — the stage keys do not depend on --corpus."""
    j, det = _run(with_detail=True)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "recall", "stages_summary"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0 < r["frac"] < 1 and len(r["kernel"]) <= 120
    for key in ("traffic", "traffic_source", "launches", "avg_launch_ms", "algorithmic_gflop_per_launch", "share_of_step"):
        assert key in r, key
    # the event pairs are recorded around every dense launch of every 4th timed step (here: step 0 of 2): 48 linears of one ragged pass
    assert r["timed_steps"] == 1 and r["launches"] == 48 and 0 < r["share_of_step"] < 1
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and 0 < len(c["sample"]) <= 160
    assert j["value"] > 0 and abs(j["value"] - 64 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-3
    rec = j["recall"]
    assert rec["gpu"] == rec["cpu_oracle"], "Recall@{1,10,100} parity with the CPU oracle"
    assert rec["rows_violating_tie_rule"] == 0 and rec["topk_ids_identical_rows"] + 8 >= rec["rows"]
    assert j["config"]["encoder_rows"] == "ragged" and j["config"]["workload"].startswith("C2/ragged")
    assert len(j["config"]["workload"]) <= 100 and "ragged encoder" in det["config_note"]
    assert "static" in r["traffic_source"] or r["traffic"] is None
    assert c["cpu_model"]
    ss = j["stages_summary"]                           # a dozen scalars of the other stages; the full object is on the #stages line
    assert len(ss) <= 40, ss
    for k_, v in ss.items():                           # positive numbers; the *_violations counts are zero; one triple of embed times
        if k_.endswith("violations"):
            assert v == 0, (k_, v)
        elif v is None:
            assert k_.startswith("c5_"), k_
        elif isinstance(v, list):
            assert len(v) == 3 and all(x > 0 for x in v), (k_, v)
        else:
            assert isinstance(v, (int, float)) and v > 0, (k_, v)
    for key in ("c3_B64_beam10_qps", "B64_beam10_decode_ms", "B64_beam10_frac_of_floor_executed", "B1_beam100_decode_ms",
                "c3_best_sustained_qps", "bf16_c2_qps", "bf16_B64_beam30_generate_ms", "sim_B32_ms", "sim_B32_frac_of_hbm_peak",
                "doc_tower_frac_of_f32_mfma_peak", "c2_prefilter_qps", "sim_B32_prefilter_ms", "sim_B1_prefilter_ms", "B64_beam10_launches"):
        assert key in ss, key
    # r06: the driver's one line carries the reference-equivalent-work number, config C5, the C3 / C5 oracle checks, the doc tower's three
    # forms and the exploratory split-bf16 step
    for key in ("c2_padded_qps", "c2_padded_linear_frac", "c3_parity_violations", "c5_qps", "c5_linear_frac", "c5_parity_violations",
                "B1_beam100_launches", "B1_beam100_frac_of_floor_executed", "doc_tower_320k_embed_s", "c2_split_bf16_qps",
                "c2_split_bf16_tie_rule_violations"):
        assert key in ss, key
    assert ss["c2_padded_qps"] < j["value"] and 0 < ss["c2_padded_linear_frac"] < 1 and 0 < ss["c5_linear_frac"] < 1
    emb = ss["doc_tower_320k_embed_s"]                # padded fp32 > ragged fp32 > ragged bf16
    assert emb[0] > emb[1] > emb[2]
    assert "traffic_stale" in j["roofline"]
    sp = det["stages"]["c2_step_split_bf16"]           # exploratory rows beside the headline: 24 bits carried (fp32-level) and 16 bits
    assert sp["terms6"]["topk_vs_fp32_step"]["rows_violating_tie_rule"] == 0 and sp["terms6"]["pooled_max_abs_diff_vs_fp32"] < 5e-5
    assert sp["terms3"]["pooled_max_abs_diff_vs_fp32"] < 2e-4 and sp["terms3"]["significand_bits_carried"] == 16
    assert "c2_split3_16bit_qps" in ss and ss["c2_split_f16x2_qps"] > 0
    assert sp["f16x2"]["topk_vs_fp32_step"]["rows_violating_tie_rule"] == 0 and sp["f16x2"]["pooled_max_abs_diff_vs_fp32"] < 5e-5
    c5 = det["stages"]["c5_two_stage"]
    assert c5["parity"]["stage1_rows_violating"] == 0 and c5["parity"]["stage2_rows_violating"] == 0 and c5["queries_per_s"] > 0
    assert det["stages"]["c3_two_stage"]["parity"]["stage2_queries"] == 64
    # r06 (accepted by the r05 verdict, #7): the headline's corpus pass runs behind the bf16 pre-filter — the top-k of the fp32 scores
    # for every input — and says so in the workload tag; the all-fp32 step is timed beside it, held to the same oracle lists
    assert "/prefilter" in j["config"]["workload"] and j["dtype"] == "f32" and "with_bf16_prefilter" not in j
    wp = j["all_fp32"]                                # reported beside the headline, never as `value`
    assert wp["value"] > 0 and wp["rows_violating_tie_rule"] == 0 and wp["recall"] == rec["gpu"]
    assert abs(ss["c2_all_fp32_qps"] - wp["value"]) < 1e-6 and abs(ss["c2_prefilter_qps"] - j["value"]) < 1e-6
    pre = det["stages"]["c2_step_all_fp32"]
    assert pre["form"] == "all_fp32" and pre["rows_violating_tie_rule"] == 0 and pre["flagged_rows"] == 0 and pre["recall"] == rec["gpu"]
    assert os.path.exists(os.path.join(ROOT, "bench_stages.json"))
    st = det["stages"]                                 # the other stages of the path, measured after the timed region
    assert abs(ss["B64_beam10_decode_ms"] - st["generate"]["B64_beam10"]["decode_ms"]) < 1e-6
    lat = st["similarity_topk_f32"]["B32"]
    assert lat["bound"] == "hbm" and 0 < lat["frac_of_hbm_peak"] < 1
    for key in ("B64_beam10", "B1_beam100"):
        g = st["generate"][key]
        assert g["generate_ms"] > g["encoder_ms"] > 0 and 0 < g["frac_of_floor_executed"] <= g["frac_of_floor"] < 1
        assert g["decode_gflop_executed"] < g["decode_gflop_without_table"]
    assert st["c3_two_stage"]["queries_per_s"] > 0 and st["bf16_mode_c2_step"]["queries_per_s"] > 0
    for key in ("c3_two_stage", "c3_two_stage_infer_sh"):          # the CPU path beside the GDR stages (SURVEY §8d "per config")
        cb = st[key]["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] >= 1 and 0 < cb["value"] < st[key]["queries_per_s"] and cb["sample"]
        assert 0 < cb["generate_s"] <= cb["total_s"]
    assert st["c3_best_sustained"]["queries_per_s"] >= st["c3_two_stage"]["pipelined_queries_per_s"]
    assert st["prefix_table"]["nodes"] > 1
    rr = {k: v for k, v in st["rerank"].items() if k != "note"}
    assert any(k.startswith("B64_cand") for k in rr) and any(k.startswith("B1_cand") for k in rr)
    assert all(0 < v["frac_of_hbm_peak"] < 1 and v["candidates"] > 0 for v in rr.values())
    assert st["c3_two_stage"]["after_generate_ms"] < st["c3_two_stage"]["ms"]
    con = st["generate_trie_constrained"]["B64_beam10"]        # constrained beams end early: the call must be shorter
    assert 0 < con["decode_ms"] < st["generate"]["B64_beam10"]["decode_ms"] and con["two_stage_queries_per_s"] > 0
    assert st["c3_two_stage_B512"]["queries_per_s"] > st["c3_two_stage"]["queries_per_s"] > 0
    assert "c3_two_stage_B2048" not in st, "the batch sweep is behind --sweep"
    assert 0 < st["doc_tower_bert_base_L128"]["frac_of_f32_mfma_peak"] < 1
    assert len(json.dumps(det).split("frac_of_floor prices")) == 2, "the long note is printed once"


def test_bench_all_fp32_similarity_form_is_still_a_headline_option():
    """--sim-prefilter off: the all-fp32 corpus pass as the headline (the r05 line), the pre-filtered step beside it."""
    j, det = _run("--sim-prefilter", "off", "--no-c5", with_detail=True)
    assert "/prefilter" not in j["config"]["workload"] and "all_fp32" not in j
    wp = j["with_bf16_prefilter"]
    assert wp["value"] > 0 and wp["rows_violating_tie_rule"] == 0 and wp["recall"] == j["recall"]["gpu"]
    assert det["stages"]["c2_step_bf16_prefilter"]["form"] == "bf16_prefilter" and "c5_two_stage" not in det["stages"]
    assert j["stages_summary"]["c5_qps"] is None


def test_bench_exploratory_encoder_form_is_labelled_as_such():
    """--encoder-form f16x2: the exploratory line (fp32 linears carried as fp16 x 2 planes) names itself in the workload tag and in dtype,
    prices its roofline against the 16-bit matrix peak, and keeps the recall parity with the CPU oracle (the similarity is the headline's)."""
    j = _run("--encoder-form", "f16x2", "--no-stages")
    assert j["dtype"] == "f16x2" and "f16x2-linears(exploratory)" in j["config"]["workload"] and len(j["config"]["workload"]) <= 140
    assert j["roofline"]["peak"] == 2500.0 and 0 < j["roofline"]["frac"] < 1 and j["roofline"]["traffic"] is None
    assert j["recall"]["gpu"] == j["recall"]["cpu_oracle"] and j["recall"]["rows_violating_tie_rule"] == 0 and j["stages_summary"] is None


def test_bench_padded_encoder_form_gives_the_same_recall():
    a, b = _run("--encoder", "padded", "--no-stages", "--prof-every", "1"), _run("--no-stages")
    assert a["roofline"]["timed_steps"] == 2 and a["roofline"]["launches"] == 96          # --prof-every 1: every step carries the events
    assert a["config"]["encoder_rows"] == "padded" and b["config"]["encoder_rows"] == "ragged"
    assert a["recall"] == b["recall"] and a["stages_summary"] is None


def test_bench_line_contract_bf16_mode():
    j = _run("--dtype", "bf16", "--no-cpu-baseline", "--no-stages")
    assert j["dtype"] == "bf16" and j["cpu_baseline"] is None and j["roofline"]["peak"] == 2500.0


def test_bench_two_stage_workloads_and_the_self_launch():
    """--workload c3 / c5 (BASELINE configs C3 / C5: the two-stage GDR path) print the same JSON contract; `--launcher` makes
    bench.py start its rank(s) itself through torch.distributed.run (what `--gpus N` does for N > 1): the step then runs the
    SHARDED stage 2 (ShardedIndex.rerank_own) over a 1-rank RCCL group."""
    j = _run("--workload", "c3", "--gpus", "1", "--launcher", "--batch", "8", "--no-cpu-baseline")
    assert j["config"]["workload"].startswith("C3/sharded") and len(j["config"]["workload"]) <= 100
    assert j["n_gpus"] == 1 and j["dtype"] == "f32" and j["config"]["beams"] == 10 and j["config"]["candidates_per_query"] == 120
    assert j["value"] > 0 and abs(j["value"] - 8 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-3
    r = j["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and r["launches_per_step"] > 100 and r["events_lost"] == 0
    k = _run("--workload", "c5", "--batch", "8")
    assert k["config"]["workload"].startswith("C5:") and k["dtype"] == "bf16" and k["config"]["beams"] == 30
    assert k["roofline"]["peak"] == 2500.0 and k["value"] > 0
    c = k["cpu_baseline"]
    assert c["kind"] == "port" and 0 < c["value"] < k["value"] and 0 < c["generate_s"] <= c["total_s"]
    p = k["parity"]              # the step's output held against the oracle's bf16 emulation inside the bench line
    assert p["stage1_rows_violating"] == 0 and p["stage2_rows_violating"] == 0 and p["stage1_queries"] == 2 and p["stage2_queries"] == 8
    assert p["stage1_ids_shared"] >= 0.95 * p["stage1_ids_total"] and 0 < p["score_gap"] <= 5e-3
    u = _run("--workload", "c3", "--batch", "8", "--constrained")
    assert u["config"]["workload"].startswith("C3/constrained:") and u["value"] > 0
    p = u["parity"]
    assert p["stage1_rows_violating"] == 0 and p["stage2_rows_violating"] == 0 and p["stage1_queries"] == 4
    assert p["stage1_ids_shared"] >= 0.95 * p["stage1_ids_total"] and p["score_gap"] <= 1e-4


def test_c5_at_the_bench_batch_512_queries_beam_30_vs_oracle():
    """BASELINE config C5 at the batch the bench line runs — 512 queries x 30 beams = 15 360 beam rows, where the bf16 linears route
    to their large-batch forms (128- / 256-row tiles, persistent kernels, device-side row counts) — on a 100 000-row bf16 corpus:
    stage 1 of 4 queries against the oracle's bf16 emulation under hypothesis_lists_match (tie tolerance = the measured score gap),
    stage 2 of all 512 queries against retrieval_ref.rerank.  The check is the bench line's own `parity` leg (bench.two_stage_parity)."""
    j = _run("--workload", "c5", "--batch", "512", "--corpus", "100000", "--parity-queries", "4", "512")
    assert j["config"]["batch_per_gpu"] == 512 and j["config"]["beams"] == 30 and j["dtype"] == "bf16"
    p = j["parity"]
    assert p["stage1_queries"] == 4 and p["stage2_queries"] == 512
    assert p["stage1_rows_violating"] == 0 and p["stage2_rows_violating"] == 0
    assert p["stage1_ids_shared"] >= 0.95 * p["stage1_ids_total"] and 0 < p["score_gap"] <= 5e-3


def test_bench_gpus_2_on_one_gpu_through_gloo():
    """`python bench.py --gpus 2` as the driver types it (no launcher, no RANK in the environment): bench.py starts its two ranks
    itself; `--backend gloo` lets them share this box's one GPU (RCCL refuses that), so the whole N > 1 path runs with real
    compute — C2 layout: row-sharded corpus, query all-gather, per-shard top-k, ONE all-to-all of the packed lists on a side
    stream, merge; C3 layout: data-parallel decode + ShardedIndex.rerank_own over two shards — and rank 0 prints one JSON line."""
    j = _run("--gpus", "2", "--backend", "gloo", "--no-stages")
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 128 and j["config"]["workload"].startswith("C4-layout")
    assert j["config"]["dist_backend"] == "gloo" and j["value"] > 0 and j["cpu_baseline"] is None
    assert abs(j["value"] - 128 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-3
    p = _run("--gpus", "2", "--backend", "gloo", "--no-stages", "--sim-prefilter", "bf16")     # every shard behind its bf16 pre-filter
    assert p["n_gpus"] == 2 and p["config"]["workload"].startswith("C4-layout/ragged/prefilter") and p["value"] > 0
    k = _run("--gpus", "2", "--backend", "gloo", "--workload", "c3", "--batch", "8")
    assert k["n_gpus"] == 2 and k["config"]["workload"].startswith("C3/sharded") and k["config"]["global_batch"] == 16
    assert k["value"] > 0 and k["config"]["candidates_per_query"] == 120


def test_bench_gpus_8_on_one_gpu_through_gloo():
    """The driver's SCALE command shape at N = 8 — `python bench.py --gpus 8` — with the ranks sharing this box's one GPU over gloo, so
    that a run on a real 8-GPU node exercises nothing for the first time except RCCL itself: C4's layout (corpus row-sharded 8 ways,
    one all-to-all of the packed per-shard top-k) and C5's (8 cluster-aligned shards of a bf16 corpus, beam 30, sharded rerank);
    exactly one JSON line <= 4 KB with n_gpus 8."""
    j = _run("--gpus", "8", "--backend", "gloo", "--batch", "16", "--no-stages")
    assert j["n_gpus"] == 8 and j["config"]["global_batch"] == 128 and j["config"]["workload"].startswith("C4-layout")
    assert j["config"]["dist_backend"] == "gloo" and j["value"] > 0 and j["scaling"] == "weak"
    k = _run("--gpus", "8", "--backend", "gloo", "--workload", "c5", "--batch", "4", "--corpus", "100000")
    assert k["n_gpus"] == 8 and k["config"]["workload"].startswith("C5/sharded") and k["config"]["global_batch"] == 32
    assert k["dtype"] == "bf16" and k["config"]["beams"] == 30 and k["value"] > 0


def test_bench_under_torchrun_one_rank_rccl():
    """The N > 1 code path (process group over RCCL, query all-gather, all-to-all of the per-shard lists, barrier + max
    over ranks) with a 1-rank group — what one GPU can exercise of it."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    for extra in ([], ["--replicated-merge"]):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                              "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                              "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "64", "--corpus", "30000",
                              "--no-cpu-baseline", "--no-stages", *extra], capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        j = json.loads(lines[0])
        assert j["n_gpus"] == 1 and j["value"] > 0 and j["cpu_baseline"] is None
