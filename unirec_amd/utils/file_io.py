"""On-disk formats of a dataset directory prepared by the reference's ETL (examples/preprocess/prepare_data.py:87-237,
297-372), read unchanged -- mirror of unirec/utils/file_io.py:51-75:

  train.pkl / valid.pkl / test.pkl   pickled pandas DataFrame, columns user_id, item_id            ('user-item', T1)
  user_history.pkl                   pickled DataFrame: user_id, item_id rows ('user-item')  or
                                     user_id, item_seq (object column of int arrays; 'user-item_seq', T5)
  data.info                          JSON: n_users, n_items, <name>_file_format, ...
"""
import json
import os
import pickle

import yaml


def load_pkl_obj(filename):
    with open(filename, "rb") as f:
        return pickle.load(f)


def load_json(filename):
    with open(filename, "r") as f:
        return json.load(f)


def load_yaml(filename):
    with open(filename, "r") as f:
        return yaml.safe_load(f)


def load_data_info(dataset_path):
    """data.info of a prepared dataset directory -> dict (n_users, n_items, *_file_format, ...)."""
    return load_json(os.path.join(dataset_path, "data.info"))
