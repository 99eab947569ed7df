"""Host-side helpers with the reference's names (utils.py): the vocoder entry point and the attention plot that
synthesize() writes next to each waveform.

  spectrogram2wav(hp, mag)      utils.py:69-97   -> ophelia_amd.vocoder (Griffin-Lim on the GPU)
  plot_alignment(hp, A, ...)    utils.py:119-153 -> {outfile}.png (needs matplotlib; a missing matplotlib is an error the
                                                    caller can see, the driver decides whether plots are wanted)
"""
import os

import numpy as np

from .vocoder import spectrogram2wav  # noqa: F401  (re-export under the reference's module name)


def plot_alignment(hp, alignment, utt_idx, t2m_epoch, dir="", outfile="", savematrix=False):
    """Image of an (encoder_steps, decoder_steps) attention matrix with a colour bar.  File name as in the reference:
    {outfile}.png if `outfile` is given, else {dir or hp.logdir}/alignment_{config_name}_utt{idx}_epoch{E}.png;
    savematrix additionally stores the matrix ({outfile}_attention.npy / ...epoch{E}.npy)."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    outdir = dir or hp.logdir
    if not os.path.exists(outdir):
        os.mkdir(outdir)
    stem = "%s/alignment_%s_utt%s_epoch%s" % (outdir, getattr(hp, "config_name", ""), utt_idx, t2m_epoch)
    fig, ax = plt.subplots()
    try:
        im = ax.imshow(np.asarray(alignment))
        fig.colorbar(im)
        ax.set_title("Cfg=%s, t2m_epoch=%s, utt=#%s" % (getattr(hp, "config_name", ""), t2m_epoch, utt_idx))
        ax.set_ylabel("Encoder timestep")
        ax.set_xlabel("Decoder timestep")
        fig.savefig((outfile + ".png") if outfile else (stem + ".png"), format="png")
    finally:
        plt.close(fig)
    if savematrix:
        np.save((outfile + "_attention.npy") if outfile else (stem + ".npy"), alignment)
