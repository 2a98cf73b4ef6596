"""Streamed training (climsim_amd/stream.py): raw timestep chunks -> device loader on a side stream -> training steps
on the main stream, double-buffered.  The result must be the training you get by materialising every chunk first and
stepping through the same permutations (float atomics in the weight gradients: 1e-3)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from test_loader_cpu import make, raw_tree  # noqa: E402,F401


@pytest.mark.parametrize("loader_on", ["main", "side", "gaps", "gaps-chain", "gaps-opt-3"])
def test_streamed_training_equals_materialised_training(raw_tree, lowres_assets, loader_on, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if loader_on.startswith("gaps-"):               # the next chunk's loader in slices beside this chunk's steps (round 5)
        parts = loader_on.split("-")
        monkeypatch.setenv("CS_STREAM_GAP", parts[1])
        if len(parts) > 2:
            monkeypatch.setenv("CS_STREAM_SLICES", parts[2])
        loader_on = "gaps"
    from climsim_amd import build
    build.build()
    from climsim_amd.loader import GpuColumnLoader
    from climsim_amd.mlp import MLPEmulator
    from climsim_amd.stream import StreamedTrainer
    root, _ = raw_tree
    du = make(lowres_assets, "pytorch", root)
    ld = GpuColumnLoader(du)
    files = du.get_filelist("train")
    raws = [ld.read_raw(f) for f in files]
    # four chunks: host arrays (pinned staging path) and device tensors (HBM-resident raw shard), T = 1 or 2 timesteps
    chunks = [(raws[0][0][None], raws[0][1][None]),
              (torch.from_numpy(raws[1][0][None]).cuda(), torch.from_numpy(raws[1][1][None]).cuda()),
              (np.stack([raws[2][0], raws[0][0]]), np.stack([raws[2][1], raws[0][1]])),
              (raws[1][0][None], raws[1][1][None])]
    B, LR = 128, 1e-3

    a = MLPEmulator(units=(128, 128), max_batch=B, seed=3)
    st = StreamedTrainer(a, ld, batch_size=B, slots=2, loader_on=loader_on, shuffle="torch")      # loader kernel on the training stream (default) / on the side stream
    out = st.fit_chunks(iter(chunks), learning_rate=LR, passes_per_chunk=2, seed=11)
    rows = sum(c[0].shape[0] * c[0].shape[2] for c in chunks)
    assert out["rows"] == 2 * rows and out["steps"] == 2 * sum(-(-c[0].shape[0] * 384 // B) for c in chunks) == a.iterations

    b = MLPEmulator(units=(128, 128), max_batch=B, seed=3)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    tot = np.zeros(2)
    for mli, mlo in chunks:
        x, y = ld.stack_raw(mli, mlo)
        for _ in range(2):
            perm = torch.randperm(x.shape[0], device="cuda", generator=gen)
            for lo in range(0, x.shape[0], B):
                tot += b.train_on_batch(x, y, LR, row_idx=perm[lo:lo + B]).cpu().numpy()
    for wa, wb in zip(a.get_weights(), b.get_weights()):
        np.testing.assert_allclose(wa, wb, rtol=0, atol=1e-3 * max(1.0, float(np.abs(wb).max())))
    assert abs(out["loss"] - tot[0] / (2 * rows * 128)) <= 1e-3 * tot[0] / (2 * rows * 128)

    with pytest.raises(ValueError):
        StreamedTrainer(a, ld, batch_size=B, slots=1)
    with pytest.raises(ValueError):
        StreamedTrainer(a, ld, batch_size=4 * B)


@pytest.mark.parametrize("carry,shuffle", [(False, "torch"), (True, "feistel")])
def test_streamed_training_at_highres_width(raw_tree, lowres_assets, carry, shuffle):
    """BASELINE config 5 shape: chunks of 21,600-column timesteps (and one ragged 21,601-column chunk) resident in HBM as
    float64 raw fields, cfg-MLP-sized batches of 8192.  Streaming (double-buffered) must equal materialising every chunk first.
    carry=False: every chunk ends in a partial batch (43,200 = 5 x 8192 + 2,240).  carry=True (the default since round 4): the
    leftover rows join the next chunk's permutation, as the reference's `.unbatch().shuffle().batch()` batches across files
    (step2_retrain.py:266-277) - whole batches only, ONE short batch at the end of the pass, every row trained on exactly once.
    shuffle: `torch.randperm` from one generator per pass (rounds 1-3) / the one-kernel keyed permutation (climsim_amd/shuffle.py)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climsim_amd.loader import GpuColumnLoader
    from climsim_amd.mlp import MLPEmulator
    from climsim_amd.stream import StreamedTrainer
    root, _ = raw_tree
    du = make(lowres_assets, "pytorch", root)
    ld = GpuColumnLoader(du)
    g = torch.Generator(device="cuda").manual_seed(5)
    sub, div, scale = du.save_norm()
    sub_d, div_d, scale_d = (torch.from_numpy(v).cuda() for v in (sub, div, scale))
    tend = ld._tend.cpu().numpy()
    chunks = []
    for T, ncol in ((2, 21600), (1, 21601), (2, 21600)):
        xn = torch.randn((T, 124, ncol), generator=g, device="cuda", dtype=torch.float64) * 0.15
        a = xn * div_d[None, :, None] + sub_d[None, :, None]                    # raw fields whose normalised form is ~N(0, 0.15^2)
        yn = torch.randn((T, 128, ncol), generator=g, device="cuda", dtype=torch.float64) * 0.05   # scaled targets ~N(0, 0.05^2)
        b = yn / scale_d[None, :, None]
        for j in np.nonzero(tend >= 0)[0]:                                      # tendency rows: mlo state = mli state + 1200 * tendency
            b[:, j] = a[:, tend[j]] + 1200.0 * b[:, j]
        chunks.append((a, b))
    B, LR = 8192, 1e-3
    m1 = MLPEmulator(units=(128, 128), max_batch=B, seed=3)
    from climsim_amd.shuffle import chunk_seed, device_permutation
    out = StreamedTrainer(m1, ld, batch_size=B, slots=2, carry_remainder=carry, shuffle=shuffle).fit_chunks(iter(chunks), learning_rate=LR, seed=4)
    rows = sum(c[0].shape[0] * c[0].shape[2] for c in chunks)
    assert out["rows"] == rows
    assert out["steps"] == (-(-rows // B) if carry else sum(-(-c[0].shape[0] * c[0].shape[2] // B) for c in chunks)) == m1.iterations
    m2 = MLPEmulator(units=(128, 128), max_batch=B, seed=3)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4)
    tot = np.zeros(2)
    cx = cy = None                                   # carried rows (carry=True): materialised here with plain torch ops
    seen = nperm = 0
    for mli, mlo in chunks:
        x, y = ld.stack_raw(mli, mlo)
        assert bool(torch.isfinite(x).all()) and bool(torch.isfinite(y).all())
        if carry and cx is not None:
            x, y = torch.cat([x, cx]), torch.cat([y, cy])
        perm = torch.randperm(x.shape[0], device="cuda", generator=gen) if shuffle == "torch" else device_permutation(x.shape[0], chunk_seed(4, nperm))
        nperm += 1
        n_train = (x.shape[0] // B) * B if carry else x.shape[0]
        for lo in range(0, n_train, B):
            tot += m2.train_on_batch(x, y, LR, row_idx=perm[lo:lo + B]).cpu().numpy()
        seen += n_train
        if carry:
            cx, cy = x[perm[n_train:]].contiguous(), y[perm[n_train:]].contiguous()
    if carry and cx is not None and cx.shape[0]:
        tot += m2.train_on_batch(cx, cy, LR).cpu().numpy()
        seen += cx.shape[0]
    assert seen == rows
    for wa, wb in zip(m1.get_weights(), m2.get_weights()):
        np.testing.assert_allclose(wa, wb, rtol=0, atol=1e-3 * max(1.0, float(np.abs(wb).max())))
    assert abs(out["loss"] - tot[0] / (rows * 128)) <= 1e-3 * tot[0] / (rows * 128)
