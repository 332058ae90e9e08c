"""SomeThing wrapper -- mirror of ape/modeling/ape_deta/ape_deta.py:20-40."""
import torch.nn as nn


class SomeThing(nn.Module):
    def __init__(self, model_vision, model_language, **kwargs):
        super().__init__(**kwargs)
        self.model_vision = model_vision
        self.model_language = model_language
        self.model_vision.set_model_language(self.model_language)
        del self.model_language

    def forward(self, batched_inputs, do_postprocess=True):
        return self.model_vision(batched_inputs, do_postprocess=do_postprocess)

    def set_eval_dataset(self, dataset_name):
        self.model_vision.set_eval_dataset(dataset_name)
