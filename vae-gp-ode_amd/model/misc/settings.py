"""Global dtype/device settings (mirrors experiments/model/misc/settings.py:5-34)."""
import numpy
import torch


class Settings:
    torch_int = torch.int32
    numpy_int = numpy.int32
    torch_float = torch.float32
    numpy_float = numpy.float32
    jitter = 1e-5

    @property
    def device(self):
        return torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')


settings = Settings()
