"""Configer with the reference's surface (lib/utils/tools/configer.py:20-240): a JSON file merged with argparse
destinations of the form 'section:key' and trailing free-form `section.key value` overrides parsed with
literal_eval; get / exists / add / update / plus_one / to_dict / clone."""
import json
import os
import sys
from ast import literal_eval
from copy import deepcopy

from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log


class Configer(object):
    def __init__(self, args_parser=None, configs=None, config_dict=None):
        self.args_dict = {}
        if config_dict is not None:
            self.params_root = config_dict
        elif configs is not None:
            self.params_root = self._load(configs)
        elif args_parser is not None:
            self.args_dict = dict(args_parser.__dict__)
            self.params_root = self._load(args_parser.configs)
            for key, value in self.args_dict.items():
                parts = key.split(':')
                if not self.exists(*parts):
                    self.add(parts, value)
                elif value is not None:
                    self.update(parts, value)
            self._handle_remaining_args(getattr(args_parser, 'REMAIN', None) or [])
        else:
            self.params_root = {}

    @staticmethod
    def _load(path):
        if not os.path.exists(path):
            Log.error('Json Path:{} not exists!'.format(path))
            sys.exit(1)
        with open(path, 'r') as f:
            return json.load(f)

    def _handle_remaining_args(self, remain):
        assert len(remain) % 2 == 0, remain
        for i in range(0, len(remain), 2):
            key, raw = remain[i], remain[i + 1]
            try:
                value = literal_eval(raw)
            except (ValueError, SyntaxError):
                value = raw
            node = self.params_root
            parts = key.split('.')
            for j, part in enumerate(parts[:-1]):
                if part not in node:
                    node[part] = dict()
                elif not isinstance(node[part], dict):
                    Log.error('Cannot set {} below {}: it is a {}.'.format('.'.join(parts[j + 1:]),
                                                                          '.'.join(parts[:j + 1]), type(node[part])))
                    sys.exit(1)
                node = node[part]
            leaf = parts[-1]
            if leaf.endswith('+'):
                target = node.get(leaf[:-1])
                if not isinstance(target, list):
                    Log.error('Cannot append to {}: it is a {}.'.format(key[:-1], type(target)))
                    sys.exit(1)
                target.append(value)
            else:
                node[leaf] = value

    def clone(self):
        return Configer(config_dict=deepcopy(self.params_root))

    def get(self, *key):
        if len(key) == 0:
            return self.params_root
        node = self.params_root
        if len(key) > 2:
            Log.error('KeyError: {}.'.format(key))
            sys.exit(1)
        for k in key:
            if not isinstance(node, dict) or k not in node:
                Log.error('KeyError: {}.'.format(key))
                sys.exit(1)
            node = node[k]
        return node

    def exists(self, *key):
        if len(key) == 1:
            return key[0] in self.params_root
        if len(key) == 2:
            sec = self.params_root.get(key[0])
            return isinstance(sec, dict) and key[1] in sec
        return False

    def add(self, key_tuple, value):
        if self.exists(*key_tuple):
            Log.error('Key: {} existed!!!'.format(key_tuple))
            sys.exit(1)
        if len(key_tuple) == 1:
            self.params_root[key_tuple[0]] = value
        elif len(key_tuple) == 2:
            self.params_root.setdefault(key_tuple[0], dict())[key_tuple[1]] = value
        else:
            Log.error('KeyError: {}.'.format(key_tuple))
            sys.exit(1)

    def update(self, key_tuple, value):
        if not self.exists(*key_tuple):
            Log.error('Key: {} not existed!!!'.format(key_tuple))
            sys.exit(1)
        if len(key_tuple) == 1:
            self.params_root[key_tuple[0]] = value
        else:
            self.params_root[key_tuple[0]][key_tuple[1]] = value

    def resume(self, config_dict):
        self.params_root = config_dict

    def plus_one(self, *key):
        if not self.exists(*key):
            Log.error('Key: {} not existed!!!'.format(key))
            sys.exit(1)
        if len(key) == 1:
            self.params_root[key[0]] += 1
        else:
            self.params_root[key[0]][key[1]] += 1

    def to_dict(self):
        return self.params_root
