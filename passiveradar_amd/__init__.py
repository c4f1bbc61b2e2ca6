"""passiveradar_amd -- MI355X-native range-Doppler core (drop-in for the hot path of
Max-Manning/passiveRadar: fast_xambg + LS/NLMS clutter filters).

Submodules mirror the reference package layout:
    passiveradar_amd.range_doppler_processing.fast_xambg
    passiveradar_amd.clutter_removal.{LS_Filter, LS_Filter_Toeplitz, LS_Filter_Multiple, NLMS_filter}
    passiveradar_amd.signal_utils.{xcorr, frequency_shift}
    passiveradar_amd.config.getConfiguration
plus ``stream`` (the block pipeline of main.py:169-194, batched and sharded over GPUs) and
``scene`` (deterministic synthetic IQ).  All compute goes through libprcore.so (HIP, gfx950);
there is no CPU fallback.
"""
__version__ = "0.1.0"
