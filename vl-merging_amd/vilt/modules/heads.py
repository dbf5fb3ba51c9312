"""Heads of the hot-path losses (reference src/vilt/modules/heads.py:8-53), GEMMs on the MFMA kernel."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import engine


class Pooler(nn.Module):
    def __init__(self, hidden_size):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)

    def forward(self, hidden_states, cls_rows=None):
        """cls_rows: hidden_states[:, 0] when the caller already holds that view (engine.feature_views)."""
        x = hidden_states[:, 0] if cls_rows is None else cls_rows
        return engine.linear(x, self.dense.weight, self.dense.bias, act="tanh")  # fp32, like tanh(x.float())


class ITMHead(nn.Module):
    def __init__(self, hidden_size):
        super().__init__()
        self.fc = nn.Linear(hidden_size, 2)

    def forward(self, x):
        return engine.linear(x, self.fc.weight, self.fc.bias)


class IFMHead(nn.Module):
    def __init__(self, hidden_size):
        super().__init__()
        self.fc = nn.Linear(hidden_size, hidden_size, bias=False)

    def forward(self, x):
        return engine.linear(x, self.fc.weight, None)


class _PredictionHeadTransform(nn.Module):
    """HF BertPredictionHeadTransform: dense -> gelu -> LayerNorm(eps=1e-12) (same parameter names)."""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.LayerNorm = nn.LayerNorm(hidden_size, eps=eps)

    def forward(self, x):
        h = engine.linear(x, self.dense.weight, self.dense.bias, gelu=True)
        return engine.layer_norm(h, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)


class MLMHead(nn.Module):
    def __init__(self, hidden_size, vocab_size):
        super().__init__()
        self.transform = _PredictionHeadTransform(hidden_size)
        self.decoder = nn.Linear(hidden_size, vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(vocab_size))

    def forward(self, x):
        return engine.linear(self.transform(x), self.decoder.weight, self.bias)


class MLPClassifier(nn.Sequential):
    """Linear -> LayerNorm -> GELU -> Linear, the VQA / NLVR2 classifier of vilt_module.py:300-327.  An nn.Sequential so
    that the parameter names (`<head>.0.weight`, `.1.weight`, `.3.weight`) are the reference's; both linears run on the
    MFMA GEMM, the [B, hidden] GELU stays in torch."""

    def __init__(self, in_features, hidden, out_features):
        super().__init__(nn.Linear(in_features, hidden), nn.LayerNorm(hidden), nn.GELU(), nn.Linear(hidden, out_features))

    def forward(self, x):
        h = engine.linear(x, self[0].weight, self[0].bias)
        h = engine.layer_norm(h, self[1].weight, self[1].bias, self[1].eps, out_f32=True)
        return engine.linear(F.gelu(h), self[3].weight, self[3].bias)
