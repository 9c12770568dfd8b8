"""Flat parameter buffer layout shared with the kernels (csrc/pinn_mlp_core.h ParamLayout).

state_dict order of the reference's DNN (01:399-419), each tensor in torch layout
[out, in] row-major, every tensor starting on a 16-byte boundary:
    W_0 b_0 ... W_{h-1} b_{h-1}  W_p b_p(+3 pad)  Wv_0 bv_0  Wv_1 bv_1  Wv_2 bv_2(+3 pad)
"""


def param_shapes(n_in, hidden, n_hidden, n_out=1):
    H = hidden
    shapes = [("layers.layer_0.weight", (H, n_in)), ("layers.layer_0.bias", (H,))]
    for l in range(1, n_hidden):
        shapes += [("layers.layer_%d.weight" % l, (H, H)), ("layers.layer_%d.bias" % l, (H,))]
    shapes += [("predict.weight", (n_out, H)), ("predict.bias", (n_out,)),
               ("var_layers.0.weight", (H // 2, H)), ("var_layers.0.bias", (H // 2,)),
               ("var_layers.3.weight", (H // 4, H // 2)), ("var_layers.3.bias", (H // 4,)),
               ("var_layers.5.weight", (n_out, H // 4)), ("var_layers.5.bias", (n_out,))]
    return shapes


def param_offsets(n_in, hidden, n_hidden, n_out=1):
    """[(name, shape, float offset)], total floats (multiple of 4)."""
    out, off = [], 0
    for name, shape in param_shapes(n_in, hidden, n_hidden, n_out):
        n = 1
        for s in shape:
            n *= s
        out.append((name, shape, off))
        off += (n + 3) // 4 * 4
    return out, off


def check_arch(layers):
    """Validate `layers` for the gfx950 kernels; returns (n_in, H, n_hidden).

    The reference's `DNN.__init__` (01:389-419) accepts ANY list [n_in, h_1, ..., h_k, n_out].  This build runs the shapes
    the reference and BASELINE.json use -- [8, H x k, 1] with one width H in {128, 256} (register-resident chain) or
    {512, 1024, 2048} (layer-by-layer kernels) and 1 <= k <= 8 -- and says so for anything else instead of running it
    slowly: unequal widths (e.g. the [8, 32, 32, 32, 1] of the reference's commented-out experiments), other widths, n_in != 8
    (the physics residuals read 8 fixed columns, 01:136-137) or n_out != 1 are a ValueError naming the restriction."""
    layers = [int(v) for v in layers]
    if len(layers) < 3:
        raise ValueError("layers must be [n_in, hidden..., n_out]")
    n_in, n_out, hid = layers[0], layers[-1], layers[1:-1]
    if n_in != 8 or n_out != 1:
        raise ValueError("the reference accepts any layers list (01:389-419); the gfx950 kernels support n_in=8 (the eight columns the "
                         "physics residuals read) and n_out=1 only (got %r)" % (layers,))
    if any(h != hid[0] for h in hid):
        raise ValueError("the reference accepts unequal hidden widths (01:399-403); the gfx950 kernels need ONE width for all hidden "
                         "layers (got %r)" % (layers,))
    if hid[0] not in (128, 256, 512, 1024, 2048):
        raise ValueError("the reference accepts any hidden width; the gfx950 kernels support 128 or 256 (register-resident MFMA chain) "
                         "and 512 / 1024 / 2048 (layer-by-layer kernels, every precision but 'fp32'); got %d" % hid[0])
    if not (1 <= len(hid) <= 8):
        raise ValueError("1..8 hidden layers supported")
    return n_in, hid[0], len(hid)
