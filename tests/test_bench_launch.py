"""`python bench.py --gpus N` without torchrun around it must start its own ranks (BASELINE.json configs[3] is driven that
way): the parent spawns `python -m torch.distributed.run ... bench.py --gpus N ...` as a child BEFORE any GPU call, relays
rank 0's single JSON line and returns the children's status.  --launch-check runs that path with the GPU work replaced
by a gloo all-reduce, so the launcher is covered on a CPU-only box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_launch_command_is_the_drivers_form():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "5"], 29555)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-5] == BENCH and cmd[-4:] == ["--gpus", "8", "--steps", "5"]


@pytest.mark.parametrize("n", [2, 3])
def test_self_launch_spawns_ranks_and_relays_one_json_line(n):
    r = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # ONE JSON line on stdout, everything else on stderr
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "n_gpus": n, "rank_sum": n * (n + 1) / 2}


def test_single_rank_needs_no_launcher():
    r = _run(["--gpus", "1", "--launch-check"])
    assert r.returncode == 0 and json.loads(r.stdout.strip()) == {"launch_check": True, "n_gpus": 1, "rank_sum": 1.0}


def test_child_failure_is_the_parents_exit_status():
    r = _run(["--gpus", "2", "--launch-check", "--workload", "cfg3"], env_extra={"HMV_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not r.stdout.strip()


def test_gpus_must_match_world_size_under_torchrun():
    r = _run(["--gpus", "4", "--launch-check"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


def test_workloads_cover_the_baseline_configs_and_the_lq_bench_config_is_warning_free():
    """bench.py names BASELINE.json configs[0..2] (cfg1 = HO3D_HandMvNet.yaml's B1 x V4 x 128^2, cfg2, cfg3); the learnable-query
    bench configuration leaves out the keys that module ignores (no UserWarning from config_from_params)."""
    import warnings
    sys.path.insert(0, ROOT)
    import bench
    from handmvnet_amd.spec import config_from_params
    assert bench.WORKLOADS["cfg1"] == ("50_paper", [1024], 4, 1, 128)
    assert bench.WORKLOADS["cfg2"][2:] == (4, 8, 256) and bench.WORKLOADS["cfg3"][2:] == (8, 32, 256)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        cfg = config_from_params(*bench.params("50_paper", [1024], 8, 32, 256, "cross_attn_learnable_query"))
    assert cfg.learnable_query


def test_forward_block_reports_step_floors_for_the_fp16_modes():
    sys.path.insert(0, ROOT)
    import bench
    fam = {"a": {"bytes": 8e9, "t_hbm": 1e-3, "t_mfma": 2e-4}, "b": {"bytes": 8e9, "t_hbm": 1e-3, "t_mfma": 3e-3}}
    f32 = bench.forward_block("f32", 157.3e12 * 0.05, 100.0, fam, 1)
    assert f32["frac_of_f32_mfma_peak"] == 0.5 and "step_floor_ms" not in f32
    f16 = bench.forward_block("f16", 2.5e15 * 1e-3, 4.0, fam, 1)       # 1 ms of MFMA work, 2 ms of HBM bytes, 4 ms measured
    assert f16["step_floor_ms"] == {"mfma": 1.0, "hbm@8TB/s": 2.0, "sum_of_per_launch_floors": 4.0} and f16["frac_of_step_floor"] == 0.5
    assert "frac_of_f32_mfma_peak" not in f16


def test_pmc_summary_names_the_round3_kernels_like_the_engine_does():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary as ps
    assert ps.family("void hmv::conv_stream_f16<2, 2, 1, 8, 4, 4, true, true, 0, false>(hmv::ConvParams)") == "conv_stream_f16<64x512,k256,res>"
    assert ps.family("void hmv::conv_stream_f16<1, 2, 8, 1, 4, 4, false, false, 0, false>(hmv::ConvParams)") == "conv_stream_f16<256x64,k256>"
    assert ps.family("void hmv::conv_stream_f16<2, 2, 2, 4, 2, 8, false, false, 0, true>(hmv::ConvParams)") == "conv_stream_f16<128x256,k128,dual>"
    assert ps.family("void hmv::conv_gemm8_f16<true, false>(hmv::ConvParams)") == "conv_gemm8_f16<256x256,1x1,dual>"
    # round 4: the 16x16x32-MFMA instantiations and their small-launch companions
    assert ps.family("void hmv::conv_gemm8_f16<false, true>(hmv::ConvParams)") == "conv_gemm8_f16<256x256,1x1,m16>"
    assert ps.family("void hmv::conv_gemm8_f16<true, true>(hmv::ConvParams)") == "conv_gemm8_f16<256x256,1x1,dual,m16>"
    assert ps.family("void hmv::conv_ht_f16<true>(hmv::ConvParams)") == "conv_ht_f16<512x128,3x3,m16>"
    assert ps.family("_ZN3hmv11conv_ht_f16ILb0EEEvNS_10ConvParamsE.kd") == "conv_ht_f16<512x128,3x3>"
    assert ps.family("void hmv::conv_gemm8p_f16<true>(hmv::ConvParams)") == "conv_gemm8_f16<256x256,1x1,dual,m16,persistent>"
    assert ps.family("_ZN3hmv15conv_gemm8p_f16ILb0EEEvNS_10ConvParamsE.kd") == "conv_gemm8_f16<256x256,1x1,m16,persistent>"
    assert ps.family("hmv::conv_htp_f16(hmv::ConvParams)") == "conv_ht_f16<512x128,3x3,m16,persistent>"
    assert ps.family("_ZN3hmv12conv_htp_f16ENS_10ConvParamsE.kd") == "conv_ht_f16<512x128,3x3,m16,persistent>"
    assert ps.family("void hmv::conv_m16_f16<64, 64, true, false>(hmv::ConvParams)") == "conv_m16_f16<64x64,taps,c32>"
    assert ps.family("void hmv::conv_m16_f16<128, 128, false, true>(hmv::ConvParams)") == "conv_m16_f16<128x128,1x1,dual>"
    assert ps.family("void hmv::gemm_x3_f16<128, 128, 4, 2>(hmv::ConvParams)") == "gemm_x3_f16<128x128>"
    assert ps.family("void hmv::conv_hs_f16<3, 3, 8, 2, 1, 4, 2, 3, false>(hmv::ConvParams)") == "conv_hs_f16<3x3,64->64>"
    assert ps.family("void hmv::conv_hs_f16<3, 3, 5, 2, 1, 4, 2, 3, true>(hmv::ConvParams)") == "conv_hs_f16<3x3,40->40,res>"
    assert ps.family("void hmv::conv_hs_f16<4, 4, 2, 2, 1, 4, 2, 4, false>(hmv::ConvParams)") == "conv_hs_f16<4x4,16->64>"
    # the chained / pooled launches (template arguments N2 and POOL)
    assert ps.family("void hmv::conv_stream_f16<2, 2, 2, 4, 2, 8, false, false, 0, true, 0>(hmv::ConvParams)") == "conv_stream_f16<128x256,k128,dual>"
    assert ps.family("void hmv::conv_stream_f16<2, 2, 2, 4, 2, 4, false, false, 0, true, 64>(hmv::ConvParams)") == "conv_stream_f16<128x256,k128,dual,+1x1:64>"
    assert ps.family("void hmv::conv_stream_f16<2, 2, 2, 4, 1, 2, true, false, 0, false, 128>(hmv::ConvParams)") == "conv_stream_f16<128x256,k64,res,+1x1:128>"
    assert ps.family("void hmv::conv_stream_f16<1, 2, 8, 1, 1, 4, false, false, 0, false, 0>(hmv::ConvParams)") == "conv_stream_f16<256x64,k64>"
    assert ps.family("void hmv::conv_hs_f16<4, 4, 2, 1, 2, 8, 1, 4, false, true>(hmv::ConvParams)") == "conv_hs_f16<4x4,16->64,+maxpool>"
    assert ps.family("void hmv::conv_hs_f16<4, 4, 2, 1, 2, 8, 1, 4, false, false>(hmv::ConvParams)") == "conv_hs_f16<4x4,16->64>"
    assert ps.family("void hmv::conv_hs_f16<3, 3, 8, 2, 1, 4, 2, 2, true, false>(hmv::ConvParams)") == "conv_hs_f16<3x3,64->64,res>"
    assert ps.family("void hmv::conv_stream_f32<2, 1, 1, 8, 2, 4, true, false, false>(hmv::ConvParams)") == "conv_stream_f32<64x256,k64,res>"
    assert ps.family("void hmv::conv_stream_f32<2, 1, 1, 8, 4, 8, false, true, false>(hmv::ConvParams)") == "conv_stream_f32<64x256,k128,dual>"
    assert ps.family("void hmv::conv_stream_f32<1, 1, 4, 2, 8, 8, false, false, false>(hmv::ConvParams)") == "conv_stream_f32<128x64,k256>"
    assert ps.family("void hmv::conv_stream_f32<2, 1, 1, 4, 8, 4, true, false, true>(hmv::ConvParams)") == "conv_stream_f32<64x128,k256,res>"
    assert ps.family("void hmv::conv_hs_stem_f32<4>(hmv::ConvParams)") == "conv_hs_stem_f32<4x4,12->64>"
    assert ps.family("void hmv::conv_igemm<float, 256, 64, 4, 2, 0, false, 32, false, false, false, false>(hmv::ConvParams)") == "conv_igemm_f32<256x64,taps>"
    assert ps.family("void hmv::conv_rds_f32<40, true, 4>(hmv::ConvParams)") == "conv_rds_f32<3x3,40->40,res>"
    assert ps.family("void hmv::conv_rds_f32<80, false, 8>(hmv::ConvParams)") == "conv_rds_f32<3x3,80->80>"
    assert ps.family("void hmv::conv_hs_f16<3, 3, 10, 2, 1, 2, 3, 3, true>(hmv::ConvParams)") == "conv_hs_f16<3x3,80->80,res>"
    assert ps.family("void hmv::conv_igemm<float, 256, 256, 2, 4, 0, false, 32, false, false, false, false>(hmv::ConvParams)") == "conv_igemm_f32<256x256,taps>"
    assert ps.family("void hmv::conv_igemm<float, 256, 128, 4, 2, 1, false, 16, false, false, false, false>(hmv::ConvParams)") == "conv_igemm_f32<256x128,k16,1x1>"
    assert ps.family("void hmv::conv_igemm<float, 128, 128, 2, 2, 2, false, 32, false, true, false, false>(hmv::ConvParams)") == "conv_igemm_f32<128x128,dense,rowsum>"
    assert ps.family("_ZN3hmv10conv_igemmIDF16_Li64ELi64ELi2ELi2ELi0ELb0ELi32ELb0ELb0ELb0ELb1EEvNS_10ConvParamsE") == "conv_igemm_f16<64x64,taps,c32>"
    assert ps.family("_ZN3hmv10conv_igemmIDF16_Li256ELi256ELi2ELi4ELi3ELb0ELi64ELb0ELb0ELb0ELb0EEvNS_10ConvParamsE") == "conv_igemm_f16<256x256,halo>"
