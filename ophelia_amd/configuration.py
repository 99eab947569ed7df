"""Config API of the hot path: counterpart of the reference's configuration.py
(load_config 60-66, Hyperparams 40-57, CONFIG_DEFAULTS 8-35).

A config is an executable Python file (the reference's config/*.cfg work unchanged:
they compute derived values such as hop_length, full_dim, and paths from __file__).
Every public, non-module top-level name becomes an attribute of a picklable
Hyperparams object; options added late in the reference's life get defaults."""
import os
import runpy
import types

# (name, default) pairs -- the same option names and default values the reference supplies
# for configs that predate them (configuration.py:8-35)
CONFIG_DEFAULTS = {
    "initialise_weights_from_existing": [],
    "update_weights": [],
    "num_threads": 8,
    "plot_attention_every_n_epochs": 0,
    "num_sentences_to_plot_attention": 0,
    "concatenate_query": True,
    "use_external_durations": False,
    "text_encoder_type": "DCTTS_standard",
    "merlin_label_dir": "",
    "merlin_lab_dim": 592,
    "bucket_data_by": "text_length",
    "history_type": "DCTTS_standard",
    "beta1": 0.9,
    "beta2": 0.999,
    "epsilon": 0.00000001,
    "decay_lr": True,
    "squash_output_t2m": True,
    "squash_output_ssrn": True,
    "store_synth_features": False,
    "turn_off_monotonic_for_synthesis": False,
    "lw_cdp": 0.0,
    "lw_ain": 0.0,
    "lw_aout": 0.0,
    "attention_guide_fa": False,
    "select_central": False,
    "MerlinTextEncWithPhoneEmbedding": False,
}


class Hyperparams(object):
    """Attribute bag built from a config namespace (picklable, unlike a module)."""

    def __init__(self, namespace):
        for key, value in dict(namespace).items():
            if key.startswith("_") or isinstance(value, types.ModuleType):
                continue
            setattr(self, key, value)

    def validate(self):
        for name, default in CONFIG_DEFAULTS.items():
            if not hasattr(self, name):
                setattr(self, name, list(default) if isinstance(default, list) else default)


def load_config(config_fname):
    config = os.path.abspath(config_fname)
    assert os.path.isfile(config), "Config file %s does not exist" % (config)
    hp = Hyperparams(runpy.run_path(config, run_name="config"))
    hp.validate()
    return hp
