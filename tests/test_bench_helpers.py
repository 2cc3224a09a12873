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
