#!/usr/bin/env python3
"""Decode the network of the reference-held fixture test/models_saved/fnn_train_result.jls into tests/golden/fnn_qtp_fixture.json.

Run in the build container only (it reads /root/reference, which does not travel to the GPU box):

    python tests/golden/make_fnn_fixture.py

The fixture is a Julia `Serialization` stream (header "7JL", format version 0x11) of an MLJ machine; the reference's own test reads
it with `machine("./models_saved/fnn_train_result.jls")` and takes
    f_fnn = fitted_params(fitted_params(fnn_machine).machine).best_fitted_params[1]        (test/computation_mpc_test.jl:52-56)
There is no Julia here, so the stream is read at byte level.  What is needed from the format (tags of Julia's Serialization.jl):
    0x34 / 0x35  OBJECT / REF_OBJECT: the type, then the fields in declaration order (a struct's field count is NOT in the stream)
    0x10         DataType: name (0x01 len bytes), module, Int32 parameter count, parameters
    0x15         Array: 0x00?, element type tag (0x0d = Float32), dims -- a 0x14 n tuple of ints for n >= 2 dimensions, one int for a
                 vector -- then the raw little-endian data, column-major
    0xdf + k     the Int64 literal k (k <= 32); 0x31 + 4 bytes: an Int64 that fits an Int32
    0x2c + 2 B   back-reference to an earlier object
Field order puts a machine's `model`, `old_model`, `fitresult`, `cache` one after the other (MLJBase.Machine), which gives the stream
this skeleton (offsets of the type names):
    Machine{DeterministicIteratedModel{DeterministicTunedModel{AdaptiveParticleSwarm, MultitargetNeuralNetworkRegressor{Fnn,..}}}}   @10
      fitresult = Machine{DeterministicTunedModel}                                            @2461   = fitted_params(fnn_machine).machine
        fitresult = Machine{MultitargetNeuralNetworkRegressor}   (the best model, retrained)  @3258
          fitresult = (Chain(fnn_input = Dense(6 => 13; bias=false), fnn_inner = Chain(Dense(13 => 13, relu)),
                             fnn_output = Dense(13 => 4; bias=false)), ...)                   @3519   = best_fitted_params[1]
          cache     = 10 x (6 x 2048) + (6 x 450) input batches, same for the 4 outputs: 20,930 training rows, then the optimiser state
        cache = (..., Machine{Resampler}) with ITS inner machine's chain                       @1698162
          trained on 7 x 2048 + 315 = 14,651 rows = the Holdout(0.7) split the tuning evaluated on -- not the fitted parameters.
So the network the reference's test builds its controllers from is the FIRST chain of the stream.  The script locates it structurally
(first Chain symbol after the third Machine type header), checks the shapes against each other and writes Float32 values widened to
Float64 (exact).  The scenario constants are those of test/computation_mpc_test.jl:35-49,73-76,94.
"""
import json
import os
import re
import struct
import sys

import numpy as np

SRC = "/root/reference/test/models_saved/fnn_train_result.jls"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fnn_qtp_fixture.json")


def read_int(b, p):
    t = b[p]
    if t >= 0xDF:
        return t - 0xDF, p + 1
    if t == 0x31:
        return struct.unpack("<i", b[p + 1:p + 5])[0], p + 5
    if t == 0x08:
        return struct.unpack("<q", b[p + 1:p + 9])[0], p + 9
    raise ValueError(f"unexpected integer tag 0x{t:02x} at {p}")


def next_f32_array(b, p, limit):
    """First Float32 array at or after p (before limit): returns (array as float64, offset after its data)."""
    while True:
        p = b.find(b"\x15\x00\x0d", p, limit)
        if p < 0:
            raise ValueError("no Float32 array found")
        q = p + 3
        try:
            if b[q] == 0x14:
                nd = b[q + 1]
                q += 2
                dims = []
                for _ in range(nd):
                    d, q = read_int(b, q)
                    dims.append(d)
            else:
                d, q = read_int(b, q)
                dims = [d]
        except ValueError:
            p += 1
            continue
        cnt = int(np.prod(dims))
        a = np.frombuffer(b[q:q + 4 * cnt], dtype="<f4").reshape(dims, order="F").astype(np.float64)
        return a, q + 4 * cnt


def main():
    b = open(SRC, "rb").read()
    assert b[:4] == b"7JL\x11", "not a Julia Serialization stream of the expected version"
    machines = [m.start() for m in re.finditer(rb"\x10\x01\x07Machine\x1f", b)]
    assert len(machines) >= 3, machines
    p = b.index(b"\x01\x05Chain", machines[2])                      # the best model's fitted chain
    names = b.index(b"\x01\x09fnn_input\x01\x09fnn_inner\x01\x0afnn_output", p)
    assert names - p < 200
    assert b.index(b"#identity", names) < b.index(b"\x15\x00\x0d", names), "fnn_input should be an identity Dense"
    lim = p + 4096
    W_in, q = next_f32_array(b, names, lim)
    W_h, q = next_f32_array(b, q, lim)
    b_h, q = next_f32_array(b, q, lim)
    W_out, q = next_f32_array(b, q, lim)
    H, nin = W_in.shape
    n = W_out.shape[0]
    assert W_h.shape == (H, H) and b_h.shape == (H,) and W_out.shape == (n, H) and nin == n + 2, (W_in.shape, W_h.shape, b_h.shape, W_out.shape)
    # the hidden activation is the builder's sigma (Fnn(neuron, layer, sigma = relu)); the Dense of fnn_inner back-references it
    assert b.index(b"#relu") < machines[1]
    m = nin - n
    # next Float32 array after the chain: the first training batch (n + m rows)
    batch0, _ = next_f32_array(b, q, q + 4096)
    assert batch0.shape[0] == nin

    def f(x, u):
        z = W_in @ np.concatenate([x, u])
        z = np.maximum(W_h @ z + b_h, 0.0)
        return W_out @ z

    xr, ur, x0 = np.full(n, 0.65), np.full(m, 1.2), np.full(n, 0.6)
    fix = {
        "source": "test/models_saved/fnn_train_result.jls (reference-held fixture), decoded by tests/golden/make_fnn_fixture.py",
        "layout": "W_in H x (n+m) no bias no activation; hidden (W_h, b_h) relu; W_out n x H no bias "
                  "(src/sub/model_modeler_implementation/fnn/mpc_modeler_implementation_fnn.jl:88-107)",
        "n": n, "m": m, "H": H, "L": 1, "activation": "relu",
        "W_in": W_in.tolist(), "W_h": [W_h.tolist()], "b_h": [b_h.tolist()], "W_out": W_out.tolist(),
        "scenario": {  # test/computation_mpc_test.jl:35-49,73-76,94
            "x_low": [0.2] * 4, "x_high": [1.36, 1.36, 1.30, 1.30], "u_low": [0.0, 0.0], "u_high": [4.0, 3.26],
            "horizon": 5, "sample_time": 5, "x_ref": xr.tolist(), "u_ref": ur.tolist(), "x0": x0.tolist(),
            "assert": "C_fnn_linear.x ~ C_fnn_nl.x atol 0.5; e_x likewise (test/computation_mpc_test.jl:152,163)"},
        "check": {"f_at_reference": f(xr, ur).tolist(), "f_at_x0": f(x0, ur).tolist(), "offsets": {"chain": p, "after": q}},
    }
    with open(OUT, "w") as fo:
        json.dump(fix, fo)
    print(f"wrote {OUT}: n={n} m={m} H={H}; f(x_ref,u_ref)={np.round(f(xr, ur), 4)}")


if __name__ == "__main__":
    sys.exit(main())
