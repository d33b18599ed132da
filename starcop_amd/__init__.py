"""starcop_amd: MI355X-native (gfx950) implementation of the STARCOP segmentation hot path.

Public surface (mirrors the reference):
  starcop_amd.model_module.ModelModule / configure_architecture / pred_classification / differences
  starcop_amd.model_setup.get_model
  starcop_amd.normalizer.DataNormalizer
  starcop_amd.padding.padded_predict / find_padding
  starcop_amd.mag1c.rmf / acrwl1mf / func_by_groups / generate_template_from_bands / get_mask_bad_bands
  starcop_amd.metrics
  starcop_amd.datamodule.ResidentTileSet / TrainLoader (HBM-resident training batches)
  starcop_amd.validation.run_validation ; starcop_amd.baselines.Mag1cBaseline / SanchezBaseline / VaronBaseline / binary_opening
All compute runs in starcop_amd/libstarcop_hip.so (include/starcop_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
