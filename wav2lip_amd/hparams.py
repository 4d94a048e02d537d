"""`hparams` singleton with the reference's attribute surface (hparams.py:15-95): attribute reads,
`set_hparam`, and `get_image_list` (hparams.py:4-13)."""
import os


def get_image_list(data_root, split):
    out = []
    with open(os.path.join("filelists", "%s.txt" % split)) as fh:
        for line in fh:
            name = line.strip()
            if " " in name:
                name = name.split()[0]
            out.append(os.path.join(data_root, name))
    return out


class HParams:
    def __init__(self, **kwargs):
        self.data = dict(kwargs)

    def __getattr__(self, key):
        data = self.__dict__.get("data", {})
        if key not in data:
            raise AttributeError("'HParams' object has no attribute %s" % key)
        return data[key]

    def set_hparam(self, key, value):
        self.data[key] = value

    def values(self):
        return dict(self.data)


_AUDIO = dict(num_mels=80, rescale=True, rescaling_max=0.9, use_lws=False, n_fft=800, hop_size=200, win_size=800,
              sample_rate=16000, frame_shift_ms=None, signal_normalization=True,
              allow_clipping_in_normalization=True, symmetric_mels=True, max_abs_value=4., preemphasize=True,
              preemphasis=0.97, min_level_db=-100, ref_level_db=20, fmin=55, fmax=7600)
_TRAIN = dict(img_size=96, fps=25, batch_size=16, initial_learning_rate=1e-4, nepochs=200000000000000000,
              num_workers=16, checkpoint_interval=3000, eval_interval=3000, save_optimizer_state=True,
              syncnet_wt=0.0, syncnet_batch_size=64, syncnet_lr=1e-4, syncnet_eval_interval=10000,
              syncnet_checkpoint_interval=10000, disc_wt=0.07, disc_initial_learning_rate=1e-4)

hparams = HParams(**_AUDIO, **_TRAIN)


def hparams_debug_string():
    vals = hparams.values()
    return "Hyperparameters:\n" + "\n".join("  %s: %s" % (k, vals[k]) for k in sorted(vals) if k != "sentences")
