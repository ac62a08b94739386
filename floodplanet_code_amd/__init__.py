"""floodplanet_code_amd: MI355X-native UNet segmentation training path for the FloodPlanet
``st_water_seg`` pipeline (drop-in for ``st_water_seg.models`` on that path only).

  floodplanet_code_amd.models      MODELS / build_model / WaterSegmentationModel / EarlyFusionModel
                                   (mirror of st_water_seg/models/__init__.py:5-20)
  floodplanet_code_amd.unet        HipUNet: reference-compatible nn.Module over libfloodunet.so
  floodplanet_code_amd.csrc        hand-written gfx950 HIP kernels + the C ABI (include/floodunet.h)
  floodplanet_code_amd.distributed one-process-per-GPU data parallelism (RCCL bucketed all-reduce)
  floodplanet_code_amd.fit         minimal trainer honouring the Lightning hook protocol of fit.py
"""
__version__ = "0.1.0"
