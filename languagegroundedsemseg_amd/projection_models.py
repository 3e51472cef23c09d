"""AttributeFittingModel -- the projection model behind the loss's latent attribute augmentation.

Same constructor, attribute names, state-dict keys (`attr_linears.<i>.weight / .bias`) and output as
/root/reference/models/projection_models.py:4-19: one nn.Linear(inputSize, outputSize) per attribute, applied to the same
rows, stacked to [n, num_attributes, outputSize].  The reference fills a `torch.cuda.FloatTensor` attribute by attribute;
here the eight products are ONE [n, C] x [C, A * C_out] GEMM on whatever device the input lives on (a checkpoint of the
reference's pretrained model loads unchanged: lib/losses/ContrastiveLanguageLoss.py:55-58)."""
import torch
import torch.nn as nn


class AttributeFittingModel(nn.Module):
    def __init__(self, inputSize, outputSize, num_attributes):
        super().__init__()
        self.input_size, self.output_size, self.num_attributes = inputSize, outputSize, num_attributes
        self.attr_linears = nn.ModuleList([nn.Linear(inputSize, outputSize) for _ in range(num_attributes)])

    def forward(self, x):
        w = torch.cat([l.weight for l in self.attr_linears], 0)          # [A * C_out, C_in]
        b = torch.cat([l.bias for l in self.attr_linears], 0)
        out = torch.nn.functional.linear(x.to(w.dtype), w, b)
        return out.view(x.shape[0], self.num_attributes, self.output_size)

    def project(self, x, attribute):
        """rows of x through ONE attribute's linear each: attribute [n] int64 -> [n, C_out] (what `forward(x)[:, a]` selects)"""
        return self.forward(x).gather(1, attribute.view(-1, 1, 1).expand(-1, 1, self.output_size)).squeeze(1)
