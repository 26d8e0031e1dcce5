"""ColorModel — the model surface the reference codec calls
(`from unified.model import model; model.ColorModel(config["model"])`,
sender/encoder/codec_pipeline.py:14,65).  The upstream model package and its
weights are not in the reference tree, so the architecture is
this build's own, frozen in DESIGN.md §MODEL; what is mirrored is the operator
surface the two pipelines use:

    model.g_a(x) -> (y, k)                 analysis, stride 1 -> 8
    model.g_s(y_hat, k=ks) -> x_hat        synthesis with per-frame top-k pruning
    model.g_s.down_conv(st) -> st          stride-2 coordinate map only
    model.entropy_model.h_a / h_s / scale_nn / eps / get_offsets
    model.entropy_model.entropy_bottleneck / gaussian_conditional
    model.update() / .eval() / .to()

Every layer is a call into libpcc_hip.so through the active Runtime.
"""
import os

import numpy as np
import torch

from . import runtime as _rt
from . import tables as _tables
from .sparse import SparseTensor, CoordSet
from .entropy import EntropyBottleneck, GaussianConditional

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def load_checkpoint(name="demo_small"):
    path = os.path.join(ASSETS, name + ".npz")
    with np.load(path) as f:
        return {k: f[k] for k in f.files}


def load_model_dir(base_path, model_name):
    """What the reference's load_model(base_path) reads (sender/encoder/codec_pipeline.py:56-72): the directory
    <base_path>/<model_name>/ with config.yaml and the weights.  Returns (config["model"], tensors).
      * weights.npz there (this build's checkpoint format: the tensor names of tools/make_checkpoint.py; the entropy
        models either as integer CDF tables or as raw parameters — CompressAI's state_dict names, tables.py — from
        which ColorModel.update() builds the tables) is loaded; config.yaml, when present, supplies the model section
        (name, channels, latent_channels, hyper_channels), which ColorModel checks against the tensors;
      * only weights.pt there — a torch state_dict of the reference's model package, which is not in its tree, so its
        key names, MinkowskiEngine kernel layouts and CompressAI table builders cannot be restated here: refused with
        that message rather than silently replaced;
      * no such directory: the seeded in-tree checkpoint (assets/<model_name>.npz)."""
    d = os.path.join(base_path or "", model_name)
    config = {"name": model_name}
    cfg = os.path.join(d, "config.yaml")
    if os.path.isfile(cfg):
        import yaml
        with open(cfg, "r") as f:
            loaded = yaml.safe_load(f) or {}
        config = dict(loaded.get("model", {}) or {})
        config.setdefault("name", model_name)
    npz = os.path.join(d, "weights.npz")
    if os.path.isfile(npz):
        with np.load(npz) as f:
            return config, {k: f[k] for k in f.files}
    if os.path.isfile(os.path.join(d, "weights.pt")):
        raise FileNotFoundError(
            f"{d}: weights.pt is a state_dict of the reference's model package (unified.model), which is not part of "
            f"its tree: this build cannot map its keys.  Export the tensors under the names of tools/make_checkpoint.py "
            f"to {npz} (numpy .npz) and it is loaded from there.")
    return config, load_checkpoint(model_name)


class _Params:
    """device-resident weights, one copy per device"""

    def __init__(self, tensors, device):
        self.host = tensors
        self.dev = {}
        for k, v in tensors.items():
            if (k.endswith(".weight") or k.endswith(".bias")) and not k.startswith("scale_nn"):
                self.dev[k] = torch.from_numpy(np.ascontiguousarray(v)).to(device)

    def wb(self, name):
        return self.dev[name + ".weight"], self.dev[name + ".bias"]


def conv3(p, name, x, relu):
    rt = x.rt
    w, b = p.wb(name)
    return SparseTensor(rt.sparse_conv(x.F, x.cs.nbr27(), w, b, relu), coordset=x.cs)


def down2(p, name, x, relu):
    rt = x.rt
    w, b = p.wb(name)
    pcs, nbr8, _ = x.cs.down()
    return SparseTensor(rt.sparse_conv(x.F, nbr8, w, b, relu), coordset=pcs)


def up2(p, name, x, relu):
    rt = x.rt
    w, b = p.wb(name)
    return SparseTensor(rt.convT_gen(x.F, w, b, relu), coordset=x.cs.up())


def linear(p, name, x, relu=False):
    w, b = p.wb(name)
    return x.rt.linear(x.F, w, b, relu)


class AnalysisTransform:
    """g_a: 3 x (conv3 + ReLU, stride-2 conv + ReLU), final conv3 -> C_y.
    Returns (y, k) with k[scale][frame] = voxel count of that frame at strides
    4, 2, 1 (coarse -> fine), consumed by g_s as the top-k sizes."""

    def __init__(self, params):
        self.p = params

    def __call__(self, x):
        p = self.p
        counts = []
        h = x
        for j in range(3):
            offs = h.cs.offsets
            counts.append([offs[i + 1] - offs[i] for i in range(h.cs.n_batch)])
            h = conv3(p, f"g_a.conv{j}", h, True)
            h = down2(p, f"g_a.down{j}", h, True)
            h.cs.set_batches(x.cs.n_batch)
        y = conv3(p, "g_a.conv3", h, False)
        k = [counts[2], counts[1], counts[0]]
        return y, k


class SynthesisTransform:
    """g_s: 3 x (generative up + ReLU, conv3 + ReLU, 1x1 occupancy logit,
    per-frame top-k prune), then a 1x1 colour head."""

    def __init__(self, params):
        self.p = params

    def down_conv(self, st):
        # only the output coordinate set is consumed (codec_parallel.py:302-305)
        pcs = st.cs.down()[0]
        return SparseTensor(None, coordset=pcs)

    def __call__(self, y_hat, k):
        p = self.p
        h = y_hat
        n_batch = h.cs.n_batch
        _ = h.cs.offsets
        for j in range(3):
            h = up2(p, f"g_s.up{j}", h, True)
            # conv3 + ReLU with the 1x1 occupancy head fused into its epilogue
            w, b = p.wb(f"g_s.conv{j}")
            hw, hb = p.wb(f"g_s.occ{j}")
            feats, logits = h.rt.sparse_conv_head(h.F, h.cs.nbr27(), w, b, True, hw, hb)
            h = SparseTensor(feats, coordset=h.cs)
            offs = h.cs.offsets
            kj = [min(int(k[j][f]), offs[f + 1] - offs[f]) for f in range(n_batch)]
            keep = h.rt.topk_prune(logits.view(-1), offs, kj)
            new_offs = [0]
            for v in kj:
                new_offs.append(new_offs[-1] + v)
            cs = h.cs.subset(keep, n_batch, new_offs)
            h = SparseTensor(h.rt.gather_rows(h.F, keep), coordset=cs)
        rgb = linear(p, "g_s.color", h)
        return SparseTensor(rgb, coordset=h.cs)


class HyperAnalysis:
    def __init__(self, params):
        self.p = params

    def __call__(self, y):
        h = conv3(self.p, "h_a.conv0", y, True)
        h = down2(self.p, "h_a.down0", h, True)
        return down2(self.p, "h_a.down1", h, False)


class HyperSynthesis:
    def __init__(self, params):
        self.p = params

    def __call__(self, z_hat):
        h = up2(self.p, "h_s.up0", z_hat, True)
        h = up2(self.p, "h_s.up1", h, True)
        return conv3(self.p, "h_s.conv0", h, False)


class ScaleNN:
    """scale_nn(q[1,2]) -> [1,C_y].  Evaluated on the host in float32 with a
    fixed operation order and no transcendental function, so encoder, decoder
    and oracle get identical bits: s = 0.5 + |relu(q W0 + b0) W1 + b1|."""

    def __init__(self, t):
        self.w0, self.b0 = t["scale_nn.l0.weight"], t["scale_nn.l0.bias"]
        self.w1, self.b1 = t["scale_nn.l1.weight"], t["scale_nn.l1.bias"]

    def __call__(self, q):
        q = np.asarray(q.detach().cpu().numpy() if isinstance(q, torch.Tensor) else q, dtype=np.float32)
        q = q.reshape(-1, 2)
        out = np.empty((q.shape[0], self.w1.shape[1]), dtype=np.float32)
        for r in range(q.shape[0]):
            h = self.b0.copy()
            for i in range(2):
                h = (h + q[r, i] * self.w0[i]).astype(np.float32)
            h = np.maximum(h, np.float32(0))
            o = self.b1.copy()
            for i in range(h.shape[0]):
                o = (o + h[i] * self.w1[i]).astype(np.float32)
            out[r] = np.float32(0.5) + np.abs(o)
        return out


class EntropyModel:
    def __init__(self, tensors, params):
        self.h_a = HyperAnalysis(params)
        self.h_s = HyperSynthesis(params)
        self.scale_nn = ScaleNN(tensors)
        self.eps = np.float32(tensors["entropy_model.eps"])
        self.offsets_ab = tensors["entropy_model.offsets_ab"].astype(np.float32)
        self.entropy_bottleneck = EntropyBottleneck(tensors, params)
        self.gaussian_conditional = GaussianConditional(tensors, params)

    def get_offsets(self, stdev, scale=None):
        """de-quantisation offset as a function of the (scaled) stdev; `scale`
        is accepted for signature parity (codec_parallel.py:405)."""
        a, b = self.offsets_ab
        return a / (b + stdev)


def model_widths(tensors):
    """(C, C_y, C_z) of a checkpoint: hidden width, latent channels, hyper-latent channels — read off the layers that fix
    them (the native graph of csrc/codec.hip takes the same tensors)"""
    c = int(tensors["g_a.conv0.weight"].shape[2])
    cy = int(tensors["g_a.conv3.weight"].shape[2])
    cz = int(tensors["h_a.down1.weight"].shape[2])
    return c, cy, cz


# layer -> (kernel volume, input width, output width) in terms of the config's widths: the architecture of DESIGN.md §2
def _layer_shapes(c, cy, cz):
    sh = {"g_a.conv0": (27, 4, c), "g_a.conv3": (27, c, cy), "h_a.conv0": (27, cy, c), "h_a.down0": (8, c, c),
          "h_a.down1": (8, c, cz), "h_s.up0": (8, cz, c), "h_s.up1": (8, c, c), "h_s.conv0": (27, c, 2 * cy)}
    for j in range(3):
        sh[f"g_a.down{j}"] = (8, c, c)
        if j:
            sh[f"g_a.conv{j}"] = (27, c, c)
        sh[f"g_s.up{j}"] = (8, cy if j == 0 else c, c)
        sh[f"g_s.conv{j}"] = (27, c, c)
    return sh


class ColorModel:
    """`model.ColorModel(config["model"])` (codec_pipeline.py:65).  config: {"name", "channels" (hidden width C),
    "latent_channels" (C_y), "hyper_channels" (C_z)} — the widths of the layer graph of DESIGN.md §2 (its depth is fixed
    by the reference's own constants: strides 8 and 32, codec_pipeline.py:308, codec_parallel.py:296-311).  Widths the
    config leaves out are read off the tensors; widths it names must be the tensors'."""

    def __init__(self, config=None, tensors=None):
        self.config = dict(config or {"name": "demo_small"})
        self.tensors = tensors if tensors is not None else load_checkpoint(self.config.get("name", "demo_small"))
        self.device = None
        self.params = None
        self.g_a = self.g_s = self.entropy_model = None
        self._check_config()

    def _check_config(self):
        c, cy, cz = model_widths(self.tensors)
        want = (self.config.get("channels", c), self.config.get("latent_channels", cy), self.config.get("hyper_channels", cz))
        if tuple(int(v) for v in want) != (c, cy, cz):
            raise ValueError(f"model config asks for widths (C, C_y, C_z) = {tuple(want)}, the weights have {(c, cy, cz)}")
        self.config.update({"channels": c, "latent_channels": cy, "hyper_channels": cz})
        for name, shape in _layer_shapes(c, cy, cz).items():
            got = tuple(self.tensors[name + ".weight"].shape)
            if got != shape:
                raise ValueError(f"layer {name}: weight {got}, the config's graph needs {shape}")
            if tuple(self.tensors[name + ".bias"].shape) != (shape[2],):
                raise ValueError(f"layer {name}: bias {tuple(self.tensors[name + '.bias'].shape)}, expected ({shape[2]},)")
        self.tensors["config.channels"] = np.array([c, cy, cz], dtype=np.int32)

    def load_state_dict(self, state):
        if state:
            self.tensors.update({k: np.asarray(v) for k, v in state.items()})
        return self

    def to(self, device):
        self.device = torch.device(device)
        self.params = _Params(self.tensors, self.device)
        self.g_a = AnalysisTransform(self.params)
        self.g_s = SynthesisTransform(self.params)
        _tables.update_tensors(self.tensors)      # a checkpoint with raw entropy parameters only: tables on first use
        self.entropy_model = EntropyModel(self.tensors, self.params)
        return self

    def update(self, force=False):
        """CompressAI's update() (codec_pipeline.py:69): builds the integer CDF tables of entropy_bottleneck and
        gaussian_conditional from the raw entropy parameters of the checkpoint (tables.py) — the ones it lacks, or all of
        them with force=True.  Returns True when tables were built.  Call before the native codec is created from
        `self.tensors` (load_model does)."""
        updated = _tables.update_tensors(self.tensors, force=force)
        if updated and self.device is not None:
            self.entropy_model = EntropyModel(self.tensors, self.params)
        return updated

    def eval(self):
        return self
