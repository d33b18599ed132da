"""``get_model(settings, experiment_name)``: what scripts/train.py:84 calls
(/root/reference/starcop/model_setup.py:5-20)."""
import os

from .model_module import ModelModule, load_weights


def get_model(settings, experiment_name=None):
    if settings.model.model_mode == "segmentation_output":
        model = ModelModule(settings)
    else:
        raise NotImplementedError(f"model_mode {settings.model.model_mode!r}: only the segmentation path is on the "
                                  "HIP hot path (regression twin is out of scope, SURVEY.md section 2 row 15)")
    if settings.model.test:
        assert experiment_name is not None, "Expermient name must be set on test or deploy mode"
        path_to_models = os.path.join(settings.model.model_folder, experiment_name, "model.pt").replace("\\", "/")
        model.load_state_dict(load_weights(path_to_models))
        print(f"Loaded model weights: {path_to_models}")
    return model
