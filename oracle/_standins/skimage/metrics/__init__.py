import numpy as np


def peak_signal_noise_ratio(image_true, image_test, *, data_range=None):
    """scikit-image's published definition (skimage/metrics/simple_metrics.py, 0.19-0.24): 10*log10(data_range^2 / mse) in
    float64; for floating-point images without an explicit data_range the dtype range (-1, 1) applies: data_range = 1 when
    image_true has no negative value, else 2."""
    image_true = np.asarray(image_true)
    image_test = np.asarray(image_test)
    if data_range is None:
        if image_true.dtype.kind == "f":
            data_range = 1.0 if image_true.min() >= 0 else 2.0
        else:
            info = np.iinfo(image_true.dtype)
            data_range = float(info.max) if image_true.min() >= 0 else float(info.max) - float(info.min)
    err = np.mean((image_true.astype(np.float64) - image_test.astype(np.float64)) ** 2, dtype=np.float64)
    with np.errstate(divide="ignore"):
        return float(10 * np.log10((data_range ** 2) / err))
