"""Constants of the reference's hot path, restated (testing/parameters.py:1,17-35; training/parameters.py:13-32)."""
WAVEFORM_SAMPLING_RATE = 8000

afp_settings = {
    "audfprint": {"density": 20, "pks-per-frame": 5, "freq-sd": 30, "shifts": 1, "samplerate": 8000,
                  "n_fft": 512, "n_hop": 256},
    "dejavu": {"samplerate": 8000, "n_fft": 512, "n_hop": 256, "fan_value": 3, "amp_min": 50, "peak_neighb_size": 10},
}

# training/parameters.py:13-30
BATCH_SIZE = 128
TRAIN_STEPS = 64
VAL_STEPS = 64
LEARNING_RATE = 1e-3
PATIENCE = 10
FACTOR = 0.1
EARLY_STOP = 20
NB_EPOCHS = 500
DURATION = 3
FACTOR_SC = 0.5      # MultiResolutionSTFTLoss factors of the Demucs branch (training/parameters.py:29-30)
FACTOR_MAG = 0.5
DEMUCS_LEARNING_RATE = 5e-4   # training/train.py:632
