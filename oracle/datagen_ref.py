"""Oracle for the inference data path in numpy.  TEST INFRASTRUCTURE.  PINNED: tests/golden/make_golden_datapath.py runs the
reference's own inference.py (datagen, chunking, main()) with stub cv2 / librosa; tests/test_golden_datapath.py compares.

  mel_chunk_starts   inference.py:231-240   (bit-exact integer arithmetic)
  datagen_batch      inference.py:133-143   (mask lower half, concat, /255. in float64) + :259 (NCHW, float32)
  frames_to_u8       inference.py:265,269   (x255., astype(uint8) truncation)
  crop_audio_window_start  wav2lip_train.py:75-84
"""
import numpy as np

MEL_STEP = 16  # inference.py:156


def mel_chunk_starts(n_mel_frames, fps):
    """start column of every 16-frame mel window; the last window is re-anchored at the end"""
    mult = 80. / fps
    starts, i = [], 0
    while 1:
        s = int(i * mult)
        if s + MEL_STEP > n_mel_frames:
            starts.append(n_mel_frames - MEL_STEP)
            break
        starts.append(s)
        i += 1
    return starts


def mel_chunks(mel, fps):
    return [mel[:, s:s + MEL_STEP] for s in mel_chunk_starts(mel.shape[1], fps)]


def datagen_batch(faces_u8, mels, img_size=96):
    """faces_u8 [B,S,S,3] uint8, mels [B,80,16] -> (img_batch float64 [B,S,S,6], mel_batch [B,80,16,1])"""
    img_batch = np.asarray(faces_u8)
    mel_batch = np.asarray(mels)
    img_masked = img_batch.copy()
    img_masked[:, img_size // 2:] = 0
    img_batch = np.concatenate((img_masked, img_batch), axis=3) / 255.
    mel_batch = np.reshape(mel_batch, [len(mel_batch), mel_batch.shape[1], mel_batch.shape[2], 1])
    return img_batch, mel_batch


def to_model_inputs(img_batch, mel_batch):
    """inference.py:259-260: NHWC float64 -> NCHW float32"""
    return (np.transpose(img_batch, (0, 3, 1, 2)).astype(np.float32),
            np.transpose(mel_batch, (0, 3, 1, 2)).astype(np.float32))


def frames_to_u8(pred_nchw):
    """pred float32 [B,3,H,W] -> uint8 [B,H,W,3]"""
    p = pred_nchw.transpose(0, 2, 3, 1) * 255.
    return p.astype(np.uint8)


def crop_audio_window_start(frame_num, fps=25):
    return int(80. * (frame_num / float(fps)))
