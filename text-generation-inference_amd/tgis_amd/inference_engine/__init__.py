"""Engine plugin loader: `tgis_amd.inference_engine.<DEPLOYMENT_FRAMEWORK>` must export `InferenceEngine`
(mirrors inference_engine/__init__.py:7-9 of the reference — drop-in boundary #3 of SURVEY.md §8b)."""
import importlib


def get_inference_engine_class(deployment_framework: str):
    module = importlib.import_module("tgis_amd.inference_engine." + deployment_framework)
    return module.InferenceEngine
