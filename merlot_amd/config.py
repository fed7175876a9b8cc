"""Mirror of utils/neat_config.py (YAML -> NeatConfig with .data/.model/.optimizer/.device/... dict attributes).

Accepts the reference's merlot.yaml verbatim; TPU-only keys (tpu_run_config, num_tpu_cores, ...) are tolerated and
ignored.  GCS / tf.gfile globbing is replaced by the local `glob` module.
"""
import argparse
import glob
import os
from copy import deepcopy

import yaml


class NeatConfig(object):
    def __init__(self):                                   # utils/neat_config.py:20-28
        self.data = {}
        self.model = {}
        self.optimizer = {}
        self.device = {}
        self.downstream = {}
        self.validate = {}
        self.misc = {}

    @classmethod
    def from_yaml(cls, config_file):                      # :31-42
        with open(config_file, 'r') as f:
            config_dict = yaml.load(f, Loader=yaml.FullLoader)
        return cls.from_dict(config_dict, orig_config_file=config_file)

    @classmethod
    def from_dict(cls, config_dict, orig_config_file=None):   # :44-101
        config = deepcopy(config_dict)
        if 'misc' not in config:
            config['misc'] = {}
        for key in ['data', 'model', 'optimizer', 'device']:
            if key not in config:
                raise ValueError("Configuration file {} is missing {}".format(orig_config_file, key))
        if 'output_dir' not in config['device']:
            raise ValueError("Missing output directory")
        config['device']['tpu_run_config'] = None          # TPU RunConfig has no MI355X meaning
        for x in ['train_file', 'val_file', 'test_file']:
            if x in config['data']:
                v_list = []
                for pattern in str(config['data'][x]).split(','):
                    v_list.extend(sorted(glob.glob(pattern)))
                config['data'][f'{x}_expanded'] = v_list
        obj = cls()
        obj.__dict__.update(config)
        return obj

    @classmethod
    def from_args(cls, help_message="NeatConfig", default_config_file=None, argv=None):   # :104-119
        parser = argparse.ArgumentParser(description=help_message)
        parser.add_argument('config_file', nargs='?', help='Where the config.yaml is located',
                            default=default_config_file, type=str)
        args = parser.parse_args(argv)
        if not args.config_file:
            raise ValueError("No config file provided!")
        if not os.path.exists(args.config_file):
            raise ValueError("Config file {} not found?".format(args.config_file))
        return cls.from_yaml(args.config_file)
