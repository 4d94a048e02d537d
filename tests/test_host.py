"""Host-side mirror of the reference API: module trees, state-dict keys, error behaviour.  No GPU."""
import numpy as np
import pytest
import torch

from oracle import models_ref
from wav2lip_amd import models as amd_models
from wav2lip_amd.hparams import hparams
from wav2lip_amd.models.wav2lip import fold_time


def test_state_dict_surface_matches_reference_counts():
    g, s, d = amd_models.Wav2Lip(), amd_models.SyncNet_color(), amd_models.Wav2Lip_disc_qual()
    assert (len(g.state_dict()), len(s.state_dict()), len(d.state_dict())) == (352, 217, 28)
    assert sum(p.numel() for p in g.parameters()) == 36298035
    assert sum(p.numel() for p in s.parameters()) == 16435072
    assert sum(p.numel() for p in d.parameters()) == 14113793
    keys = g.state_dict().keys()
    for k in ("face_encoder_blocks.0.0.conv_block.0.weight", "face_decoder_blocks.6.2.conv_block.1.running_mean",
              "audio_encoder.12.conv_block.1.num_batches_tracked", "output_block.1.weight", "output_block.1.bias"):
        assert k in keys
    assert isinstance(g.face_encoder_blocks, torch.nn.ModuleList) and len(g.face_encoder_blocks) == 7
    assert len(g.audio_encoder) == 13 and len(g.face_decoder_blocks) == 7 and len(g.output_block) == 3
    assert d.label_noise == .0


def test_module_geometry_agrees_with_the_oracle_tables():
    """two independent restatements of the layer stacks (product tables vs oracle strings) must agree"""
    def geoms(seq):
        out = []
        for b in seq:
            c = b.conv_block[0]
            out.append(dict(stride=tuple(c.stride), padding=c.padding[0], residual=b.residual,
                            transposed=isinstance(c, torch.nn.ConvTranspose2d),
                            output_padding=c.output_padding[0] if isinstance(c, torch.nn.ConvTranspose2d) else 0))
        return out
    g, s, d = amd_models.Wav2Lip(), amd_models.SyncNet_color(), amd_models.Wav2Lip_disc_qual()
    for blk, ref in zip(g.face_encoder_blocks, models_ref.GEN_FACE_ENC):
        assert geoms(blk) == [models_ref.parse_geom(x) for x in ref]
    for blk, ref in zip(g.face_decoder_blocks, models_ref.GEN_FACE_DEC):
        assert geoms(blk) == [models_ref.parse_geom(x) for x in ref]
    assert geoms(g.audio_encoder) == [models_ref.parse_geom(x) for x in models_ref.GEN_AUDIO_ENC]
    assert geoms(s.face_encoder) == [models_ref.parse_geom(x) for x in models_ref.SYNC_FACE_ENC]
    assert geoms(s.audio_encoder) == [models_ref.parse_geom(x) for x in models_ref.SYNC_AUDIO_ENC]
    for blk, ref in zip(d.face_encoder_blocks, models_ref.DISC_ENC):
        assert geoms(blk) == [models_ref.parse_geom(x) for x in ref]


def test_no_cpu_fallback():
    g = amd_models.Wav2Lip().eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        g(torch.zeros(1, 1, 80, 16), torch.zeros(1, 6, 96, 96))
    s = amd_models.SyncNet_color().eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        s(torch.zeros(1, 1, 80, 16), torch.zeros(1, 15, 48, 96))


def test_fold_time_is_t_major_like_the_reference():
    a = torch.randn(3, 5, 1, 80, 16)
    f = torch.randn(3, 6, 5, 8, 8)
    fa, ff = fold_time(a, f)
    assert torch.equal(fa, torch.cat([a[:, i] for i in range(5)], dim=0))
    assert torch.equal(ff, torch.cat([f[:, :, i] for i in range(5)], dim=0))


def test_hparams_surface():
    assert hparams.n_fft == 800 and hparams.hop_size == 200 and hparams.num_mels == 80 and hparams.fps == 25
    hparams.set_hparam("syncnet_wt", 0.01)
    assert hparams.syncnet_wt == 0.01
    hparams.set_hparam("syncnet_wt", 0.0)
    with pytest.raises(AttributeError):
        hparams.nonexistent


def test_resize_oracle_known_answers():
    """cv2.resize(INTER_LINEAR, uint8) restatement (oracle/resize_ref.py): identities and hand-computed fixed-point values"""
    import numpy as np
    from oracle import resize_ref as R
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (50, 70, 3), dtype=np.uint8)
    assert np.array_equal(R.resize_linear_u8(a, (70, 50)), a)                       # same size: copy
    area = R.resize_linear_u8(a, (35, 25)).astype(int)                               # exact 2x: (sum of 4 + 2) >> 2
    s = a.astype(int)
    assert np.array_equal(area, (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2)
    for d in ((96, 96), (10, 7), (200, 150)):
        assert (R.resize_linear_u8(np.full((33, 41, 3), 137, np.uint8), d) == 137).all()
    # 1x2 -> 1x4 by hand: fx = (dx+.5)*.5-.5 = -.25, .25, .75, 1.25 -> (0,0) (0,.25) (0,.75) (1,0 clamped)
    src = np.array([[[0, 0, 0], [200, 200, 200]]], dtype=np.uint8)
    out = R.resize_linear_u8(src, (4, 1))[0, :, 0].tolist()
    assert out == [0, 50, 150, 200]
    # the paste touches only the box
    frame = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    keep = frame.copy()
    R.resize_paste(frame, rng.integers(0, 256, (96, 96, 3), dtype=np.uint8), (5, 25, 10, 50))
    mask = np.ones((40, 60), bool)
    mask[5:25, 10:50] = False
    assert np.array_equal(frame[mask], keep[mask]) and not np.array_equal(frame, keep)


def test_validate_boxes():
    import pytest
    from wav2lip_amd.inference import validate_boxes
    assert validate_boxes([(0, 96, 0, 96), [1, 2, 3, 4]], 100, 100) == [(0, 96, 0, 96), (1, 2, 3, 4)]
    for bad in ((0, 0, 0, 5), (-1, 5, 0, 5), (0, 101, 0, 5), (0, 5, 7, 7), (0, 5, 0, 101)):
        with pytest.raises(ValueError):
            validate_boxes([bad], 100, 100)


def test_checkpoint_format_round_trip_with_module_prefix(tmp_path):
    """a DataParallel-style file (`module.` keys, the released weights' format) loads into the mirrored module; files we
    write carry exactly the reference's four entries"""
    import torch
    from wav2lip_amd import synthetic as synth
    from wav2lip_amd import checkpoint, models
    G = models.Wav2Lip()
    sd = synth.synthetic_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=3)
    p = tmp_path / "wav2lip_gan_like.pth"
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}, "optimizer": None, "global_step": 7, "global_epoch": 2}, p)
    G2, step, epoch = checkpoint.load_checkpoint(str(p), models.Wav2Lip())
    assert (step, epoch) == (7, 2)
    assert all(torch.equal(v, sd[k]) for k, v in G2.state_dict().items())
    out = checkpoint.save_checkpoint(G2, None, 11, str(tmp_path), 3, prefix="disc_")
    assert out.endswith("disc_checkpoint_step000000011.pth")
    blob = torch.load(out, weights_only=False)
    assert sorted(blob) == ["global_epoch", "global_step", "optimizer", "state_dict"] and blob["global_step"] == 11
    assert list(blob["state_dict"]) == list(sd)


def test_s3fd_oracle_matches_reference_golden_and_host_logic():
    """oracle/s3fd_ref.py against the fixture frozen from the real reference detector; integer smoothing semantics"""
    import os
    import sys
    import numpy as np
    import torch
    from conftest import ROOT
    from oracle import s3fd_ref
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_s3fd import images, seeded_state_dict
    gold = np.load(os.path.join(ROOT, "tests", "golden", "golden_s3fd_v1.npz"))
    sd = seeded_state_dict()
    torch.set_num_threads(8)
    outs = s3fd_ref.s3fd_forward(sd, s3fd_ref.preprocess(images()))
    for i, o in enumerate(outs):
        assert np.abs(o.numpy() - gold["out%d" % i]).max() <= 1e-5 * max(1.0, np.abs(gold["out%d" % i]).max())
    dets = s3fd_ref.detections(s3fd_ref.dense_boxes(outs))
    assert [len(d) for d in dets] == gold["n_kept"].tolist()
    assert [tuple(r) for r in s3fd_ref.rects(dets)] == [tuple(r) for r in gold["rects"].tolist()]
    # inference.py:59-66 on the integer array of :101: means are truncated when written back, and later windows read them
    b = np.array([[10, 10, 20, 20], [11, 10, 22, 21], [13, 11, 25, 22], [14, 13, 27, 23], [16, 14, 28, 25], [19, 15, 30, 27]])
    sm = s3fd_ref.get_smoothened_boxes(b.copy(), T=5)
    assert sm.dtype == b.dtype and sm[0].tolist() == [12, 11, 24, 22]          # mean 12.8, 11.6, 24.4, 22.2 truncated
    from wav2lip_amd.inference import get_smoothened_boxes
    assert np.array_equal(get_smoothened_boxes(b.copy(), 5), sm)
    from wav2lip_amd import face_detection as fd
    assert list(fd.s3fd().state_dict()) == list(sd) or sorted(fd.s3fd().state_dict()) == sorted(sd)
    import pytest
    with pytest.raises((FileNotFoundError, RuntimeError)):
        fd.FaceAlignment(fd.LandmarksType._2D, device="cuda")


# ---------------------------------------------------------------- output container (wav2lip_amd/container.py)
def _riff_tree(buf, start, end, depth=0):
    """independent walk of a RIFF byte string: [(depth, fourcc, list kind or None, payload offset, size)]; asserts alignment"""
    import struct
    out, pos = [], start
    while pos < end:
        assert pos % 2 == 0 and pos + 8 <= end
        cc, size = buf[pos:pos + 4], struct.unpack_from("<I", buf, pos + 4)[0]
        assert pos + 8 + size <= end + 1
        if cc in (b"RIFF", b"LIST"):
            kind = buf[pos + 8:pos + 12]
            out.append((depth, cc, kind, pos + 12, size - 4))
            if kind != b"movi":
                out += _riff_tree(buf, pos + 12, pos + 8 + size, depth + 1)
        else:
            out.append((depth, cc, None, pos + 8, size))
        pos += 8 + size + (size & 1)
    assert pos == end or pos == end + 1
    return out


@pytest.mark.parametrize("w,h,fps,channels", [(96, 96, 25, 1), (97, 33, 29.97, 2), (6, 5, 12.5, 0)])
def test_avi_writer_round_trip_and_structure(tmp_path, w, h, fps, channels):
    import struct
    from wav2lip_amd import container
    r = np.random.default_rng(w)
    T = 7
    frames = r.integers(0, 256, (T, h, w, 3), dtype=np.uint8)
    sr = 16000
    n_audio = int(T / fps * sr) + 1234                       # audio a little longer than the video: the tail is kept
    audio = r.integers(-32768, 32768, (n_audio, channels), dtype=np.int16) if channels else None
    path = str(tmp_path / "out.avi")
    out = container.AviWriter(path, fps, (w, h), audio=audio, audio_sr=sr)
    assert out.isOpened()
    for f in frames:
        out.write(f)
    with pytest.raises(ValueError, match="frame must be"):
        out.write(np.zeros((h + 1, w, 3), np.uint8))
    out.release()
    out.release()                                            # idempotent, like cv2's
    with pytest.raises(ValueError, match="after release"):
        out.write(frames[0])
    got = container.read_avi(path)
    assert np.array_equal(got["frames"], frames) and abs(got["fps"] - fps) < 1e-3
    if channels:
        assert got["audio_sr"] == sr and np.array_equal(got["audio"], audio)
    else:
        assert got["audio"] is None
    # structure, walked independently of the reader
    buf = open(path, "rb").read()
    assert buf[:4] == b"RIFF" and buf[8:12] == b"AVI " and struct.unpack_from("<I", buf, 4)[0] == len(buf) - 8
    tree = _riff_tree(buf, 12, len(buf))
    names = [(d, cc if kind is None else kind) for d, cc, kind, _, _ in tree]
    assert names[:5] == [(0, b"hdrl"), (1, b"avih"), (1, b"strl"), (2, b"strh"), (2, b"strf")]
    assert names[-2:] == [(0, b"movi"), (0, b"idx1")]
    avih = [t for t in tree if t[1] == b"avih"][0]
    usec, _, _, flags, total, _, nstreams, bufsize, aw, ah = struct.unpack_from("<10I", buf, avih[3])
    assert (total, aw, ah, nstreams) == (T, w, h, 2 if channels else 1) and flags & 0x10 and abs(usec - 1e6 / fps) <= 1
    stride = (w * 3 + 3) // 4 * 4
    assert bufsize == stride * h
    movi = [t for t in tree if t[2] == b"movi"][0]
    idx = [t for t in tree if t[1] == b"idx1"][0]
    entries = [struct.unpack_from("<4sIII", buf, idx[3] + 16 * i) for i in range(idx[4] // 16)]
    assert [e[0] for e in entries].count(b"00db") == T
    for cc, flags, off, size in entries:                     # every index entry points at its chunk header inside 'movi'
        pos = movi[3] - 4 + off
        assert buf[pos:pos + 4] == cc and struct.unpack_from("<I", buf, pos + 4)[0] == size and flags & 0x10
    if channels:
        assert sum(e[3] for e in entries if e[0] == b"01wb") == n_audio * 2 * channels
        assert entries[0][0] == b"00db" and entries[1][0] == b"01wb"      # interleaved frame by frame
    # bottom-up DIB rows: the first stored row of frame 0 is the LAST image row
    first = movi[3] + 8
    assert buf[first:first + w * 3] == frames[0, h - 1].tobytes()


def test_avi_mux_replaces_the_ffmpeg_step(tmp_path):
    from scipy.io import wavfile
    from wav2lip_amd import container
    r = np.random.default_rng(3)
    frames = r.integers(0, 256, (5, 20, 30, 3), dtype=np.uint8)
    pcm = r.integers(-3000, 3000, 3200, dtype=np.int16)
    wavfile.write(str(tmp_path / "a.wav"), 16000, pcm)
    container.write_avi(str(tmp_path / "v.avi"), frames, 25)
    container.mux(str(tmp_path / "a.wav"), str(tmp_path / "v.avi"), str(tmp_path / "result.avi"))
    got = container.read_avi(str(tmp_path / "result.avi"))
    assert np.array_equal(got["frames"], frames) and np.array_equal(got["audio"][:, 0], pcm) and got["fps"] == 25.0
    with pytest.raises(ValueError, match="not a RIFF AVI"):
        container.read_avi(str(tmp_path / "a.wav"))
    with pytest.raises(ValueError, match="no frames"):
        container.write_avi(str(tmp_path / "e.avi"), [], 25)


def test_write_result_takes_the_loop_output_and_the_driving_wav(tmp_path):
    """inference.py:256-277 for in-memory frames: frames + the driving audio in one file; float / wide formats stored as PCM16"""
    from scipy.io import wavfile
    from wav2lip_amd import container, inference
    r = np.random.default_rng(9)
    frames = [r.integers(0, 256, (24, 40, 3), dtype=np.uint8) for _ in range(4)]
    wavfile.write(str(tmp_path / "f32.wav"), 22050, r.uniform(-1, 1, 1000).astype(np.float32))
    out = inference.write_result(str(tmp_path / "res.avi"), frames, 25.0, str(tmp_path / "f32.wav"))
    got = container.read_avi(out)
    assert np.array_equal(got["frames"], np.stack(frames)) and got["audio_sr"] == 22050 and got["audio"].shape == (1000, 1)
    assert np.array_equal(container.read_avi(inference.write_result(str(tmp_path / "mute.avi"), frames, 25.0))["frames"], np.stack(frames))


def test_no_cpu_fallback_for_resampling_and_the_clip_store():
    """the product path fails loudly without a HIP device instead of routing through a CPU implementation"""
    if torch.cuda.is_available():
        pytest.skip("CPU-container check")
    from wav2lip_amd import audio, data
    x = np.zeros(4000, np.float32)
    assert audio.resample(x, 16000, 16000) is not None            # same rate: nothing to compute
    with pytest.raises(RuntimeError, match="HIP device"):
        audio.resample(x, 44100, 16000)
    with pytest.raises(RuntimeError, match="HIP device"):
        data.ClipStore("cpu")


def test_three_bf16_pieces_carry_an_fp32_value_and_six_products_carry_the_product():
    """the arithmetic of the split-operand implicit GEMM (csrc/conv_igemm.hip, NP = 3) restated on CPU (tools/split_bf16_accuracy.py):
    the three pieces sum back to the fp32 value exactly; against an fp64 contraction of K = 2304 the six kept piece products are not
    worse than an fp32 accumulation in MFMA-sized chunks, three products (i + j <= 1) are several times worse, nine buy nothing"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("split_acc", os.path.join(root, "tools", "split_bf16_accuracy.py"))
    sa = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sa)
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(4096) * np.exp(rng.uniform(-20, 20, 4096))).astype(np.float32)
    p0, p1, p2 = sa.split3(x)
    for p in (p0, p1, p2):      # every piece is a bf16 value: its low 16 bits are zero
        assert not (p.view(np.uint32) & 0xFFFF).any()
    assert np.array_equal((p0.astype(np.float64) + p1 + p2).astype(np.float32), x)
    M, K, N = 64, 9 * 256, 32
    a = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    b = (rng.standard_normal((K, N)) * 0.03).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    pa, pb = sa.split3(a), sa.split3(b)

    def run(pairs):
        acc = np.zeros((M, N), np.float32)
        for k in range(0, K, 16):           # one accumulator, smallest pieces first inside each K-chunk: the kernel's order
            for i, j in pairs:
                acc = (acc + (pa[i][:, k:k + 16].astype(np.float64) @ pb[j][k:k + 16].astype(np.float64)).astype(np.float32))
        return np.sqrt(((acc - ref) ** 2).mean())

    six = [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]
    e32 = np.sqrt(((sa.mm32(a, b, 2) - ref) ** 2).mean())
    e6 = run(six)
    e3 = run([(1, 0), (0, 1), (0, 0)])
    e9 = run([(2, 2), (2, 1), (1, 2)] + six)
    assert e6 <= 1.2 * e32, (e6, e32)
    assert e3 >= 4 * e6, (e3, e6)
    assert abs(e9 - e6) <= 0.05 * e6, (e9, e6)


def test_lds_layouts_of_the_operand_tiles_are_conflict_free_where_the_design_says_so():
    """tools/lds_conflicts.py (lane groups and bank maps of the MI355X guide's LDS table) on the two K-major tile layouts of
    conv_igemm.hip: the padded 80-byte rows of the one-plane kernel read conflict-free and store 2-way conflicted; the swizzled
    64-byte rows of the split-operand kernel read AND store conflict-free; unswizzled 64-byte rows would read 4-way conflicted"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lds_conflicts", os.path.join(root, "tools", "lds_conflicts.py"))
    lc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lc)
    assert lc.cycles("ds_read_b32", lambda l: 4 * l) == 2 and lc.cycles("ds_read_b32", lambda l: 128 * l) == 64   # linear / one bank
    assert lc.cycles("ds_read_b128", lambda l: 0) == 4                                                             # broadcast
    pad = lc.rows_layout(80, False)
    assert pad == {"fragment ds_read_b128": 4, "staging ds_write_b64": 8, "staging ds_write_b128": 16}
    sw = lc.rows_layout(64, True)
    assert sw == {"fragment ds_read_b128": 4, "staging ds_write_b64": 4, "staging ds_write_b128": 8}
    assert lc.rows_layout(64, False)["fragment ds_read_b128"] == 16


def test_face_detect_front_end_host_logic_matches_the_oracle():
    """inference.py:68-104 without a GPU: rects from a stub detector -> pads, clipping, in-place integer smoothing, crops,
    the halving retry on RuntimeError and the "no face" ValueError, against oracle/s3fd_ref.get_smoothened_boxes"""
    from oracle import s3fd_ref
    from wav2lip_amd import inference as inf

    class Detector:
        def __init__(self, rects, fail_above=None):
            self.rects, self.fail_above, self.sizes = rects, fail_above, []

        def get_detections_for_batch(self, imgs):
            self.sizes.append(len(imgs))
            if self.fail_above is not None and len(imgs) > self.fail_above:
                raise RuntimeError("out of memory")
            lo = sum(s for s in self.sizes[:-1] if self.fail_above is None or s <= self.fail_above)
            return self.rects[lo:lo + len(imgs)]

    rng = np.random.default_rng(0)
    for n in (1, 3, 4, 5, 9):
        imgs = [rng.integers(0, 255, (96, 128, 3), dtype=np.uint8) for _ in range(n)]
        rects = [(int(rng.integers(0, 60)), int(rng.integers(0, 40)), int(rng.integers(61, 140)), int(rng.integers(41, 110)))
                 for _ in range(n)]
        res = inf.face_detect(imgs, Detector(rects), pads=(3, 10, 5, 7), nosmooth=False, batch_size=2)
        boxes = np.array([[max(0, x1 - 5), max(0, y1 - 3), min(128, x2 + 7), min(96, y2 + 10)] for x1, y1, x2, y2 in rects])
        boxes = s3fd_ref.get_smoothened_boxes(boxes, T=5)
        for (crop, (y1, y2, x1, x2)), b, f in zip(res, boxes, imgs):
            assert (x1, y1, x2, y2) == tuple(int(v) for v in b)
            assert np.array_equal(crop, f[y1:y2, x1:x2])
    # negative pads and a rect that leaves the frame: only the near edges are clipped at 0 and the far edges at the frame size
    # (inference.py:91-98) - a far edge pushed below 0 stays negative, a near edge beyond the frame stays there
    odd = [(10, 20, 50, 4), (150, 30, 160, 60)]
    res = inf.face_detect(imgs[:2], Detector(odd), pads=(-2, -10, -4, -3), nosmooth=True, batch_size=2)
    assert [r[1] for r in res] == [(22, -6, 14, 47), (32, 50, 154, 128)]
    assert res[0][0].shape[0] == len(imgs[0][22:-6]) and res[1][0].size == 0      # numpy slicing of such boxes, as in the reference
    det = Detector(rects, fail_above=2)                       # batches of 8 and 4 fail, 2 works: 9 frames in 5 calls
    res = inf.face_detect(imgs, det, pads=(0, 0, 0, 0), nosmooth=True, batch_size=8)
    assert det.sizes == [8, 4, 2, 2, 2, 2, 1] and [r[1] for r in res] == [(y1, min(96, y2), x1, min(128, x2)) for x1, y1, x2, y2 in rects]
    with pytest.raises(RuntimeError, match="Image too big"):
        inf.face_detect(imgs, Detector(rects, fail_above=0), pads=(0, 0, 0, 0), nosmooth=True, batch_size=2)
    with pytest.raises(ValueError, match="Face not detected"):
        inf.face_detect(imgs[:2], Detector([rects[0], None]), pads=(0, 0, 0, 0), nosmooth=True, batch_size=2)


def test_host_nms_pass_matches_the_oracle():
    """the overflow route of s3fd.nms_batch (more rows above the gate than the device pass holds): bbox.py:44-64 in numpy, float32
    operation order - clustered boxes, a negative score, a degenerate box - against oracle/s3fd_ref.nms"""
    from oracle import s3fd_ref
    from wav2lip_amd.face_detection.s3fd import _nms_host
    rng = np.random.default_rng(5)
    for n, thresh in ((1, 0.3), (60, 0.3), (900, 0.5), (300, 0.0)):
        c = np.stack([rng.choice([40.0, 90.0, 200.0], n) + rng.normal(0, 6, n), rng.choice([30.0, 120.0], n) + rng.normal(0, 6, n)], 1)
        wh = rng.uniform(8, 60, (n, 2))
        d = np.concatenate([c - wh / 2, c + wh / 2, ((rng.permutation(n) + 0.5) / n)[:, None]], 1).astype(np.float32)
        if n > 1:
            d[0, 4] = -0.25
        assert _nms_host(d, thresh) == [int(i) for i in s3fd_ref.nms(d, thresh)], (n, thresh)
    z = np.array([[5, 5, 4, 4, 0.9], [50, 50, 49, 49, 0.8], [5, 5, 30, 30, 0.7]], np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        assert _nms_host(z, 0.3) == [int(i) for i in s3fd_ref.nms(z, 0.3)]
