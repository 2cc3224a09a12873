"""CPU: the measurement helpers of bench.py that need no GPU (the legs themselves are exercised on the GPU box by the driver's own bench run)."""
import bench


def test_cpu_model_and_host_threads():
    assert isinstance(bench.cpu_model(), str) and bench.cpu_model()
    assert 1 <= bench.host_threads(0) <= 64 and bench.host_threads(3) == 3


def test_power_sampler_without_a_device_reports_nothing():
    with bench.PowerSampler(0, period=0.005) as ps:         # no GPU here: no hwmon directory -> no thread, no samples
        pass
    assert ps.report() is None


def test_roofline_groups_by_kernel_and_reads_the_grid_suffix():
    rows = [dict(layer="vae:a", kernel="k_conv3x3_halo_w<16,128,2,2,true,1> f16+fp6 grid 1048576", flops=171.8e9, ms=0.30),
            dict(layer="vae:b", kernel="k_conv3x3_halo_w<16,128,2,2,true,1> f16+fp6 grid 524288 +gn", flops=171.8e9, ms=0.25),
            dict(layer="unet:c", kernel="k_conv_igemm<64,64,2,2,true,64>", flops=10e9, ms=0.05),
            dict(layer="unet:n", kernel="k_layernorm", flops=0.0, ms=0.4)]
    rf, by = bench.roofline(rows, "bf16x3", only_mfma=True)
    assert rf["kernel"].startswith("k_conv3x3_halo_w") and rf["launches_per_step"] == 2 and rf["launch_grids"] == [524288, 1048576]
    assert rf["mfma_passes_per_product"] == 1.5 and abs(rf["peak"] - 2500 / 1.5) < 0.1
    tf = 2 * 171.8e9 / 0.55e-3 / 1e12
    assert abs(rf["achieved"] - tf) < 0.5 and abs(rf["frac_of_dense_f16_peak"] - tf / 2500) < 1e-3 and abs(rf["frac"] - tf / (2500 / 1.5)) < 1e-3
    assert "k_layernorm" not in by                           # only_mfma: the dominant-kernel pick ignores non-conv kernels


def _representative_full_line():
    """Round 4's own 24 KB line (committed under profiles/): the one the driver could not parse."""
    import json, os
    return json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_bench_line.json")))


def test_compact_line_fits_the_driver_tail_and_keeps_the_contract():
    """the ONE stdout line stays under 6 KB and every contract key, `roofline` and `cpu_baseline` survive json.loads."""
    import json
    full = _representative_full_line()
    assert len(json.dumps(full)) > 20000
    text = json.dumps(bench.compact_line(full))
    assert len(text) < bench.COMPACT_LIMIT and "\n" not in text
    got = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in got, k
        assert got[k] == full[k] or k in ("config", "roofline", "cpu_baseline")
    assert got["config"]["workload"].startswith("MuseTalk step 256x256") and "model" not in got["config"]
    rf = got["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches_per_step", "avg_launch_us", "alg_gflop_per_launch", "frac_of_dense_f16_peak",
              "socket_w", "cap_w", "sclk_mhz", "step_socket_w", "step_sclk_mhz"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and all(not isinstance(v, str) or len(v) < 64 for v in rf.values())
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in got["cpu_baseline"], k
    assert got["wav2lip"]["ms_per_step"] == full["wav2lip"]["ms_per_step"] and got["ernerf"]["value"] == full["ernerf"]["value"]
    assert got["sessions"]["sessions_per_gpu_at_25fps_end_to_end"] == 22 and got["node"]["host_cores_for_8_ranks_at_capacity"] == 15.1


def test_compact_line_size_guard_drops_optional_summaries_not_contract_keys():
    import json
    full = _representative_full_line()
    full["wav2lip"]["roofline"]["kernel"] = "k" * 3000                     # something grows again: the guard sheds optional legs, never the contract
    full["ernerf"]["roofline"]["kernel"] = "e" * 3000
    got = bench.compact_line(full)
    assert len(json.dumps(got)) <= bench.COMPACT_LIMIT
    assert "roofline" in got and "cpu_baseline" in got and got["value"] == full["value"]


def test_emit_prints_one_line_and_writes_the_detail_file(tmp_path, monkeypatch, capsys):
    import json, os
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = _representative_full_line()
    bench.emit(full)
    out = capsys.readouterr().out
    assert out.count("\n") == 1 and json.loads(out)["value"] == full["value"]
    assert json.load(open(os.path.join(str(tmp_path), bench.DETAIL_NAME)))["multi_session"] == full["multi_session"]
