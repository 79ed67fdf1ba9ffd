"""Directory-listing helpers with the reference's ordering (util/file.py:14-61): natural
("alphanum") sort, so cloud_bin_2 precedes cloud_bin_10 and output naming matches."""
import os
import re


def ensure_dir(path):
    os.makedirs(path, mode=0o755, exist_ok=True)


def sorted_alphanum(items):
    def key(s):
        return [int(c) if c.isdigit() else c for c in re.split("([0-9]+)", s)]
    return sorted(items, key=key)


def get_file_list(path, extension=None):
    names = [os.path.join(path, f) for f in os.listdir(path) if os.path.isfile(os.path.join(path, f))]
    if extension is not None:
        names = [f for f in names if os.path.splitext(f)[1] == extension]
    return sorted_alphanum(names)


def get_folder_list(path):
    return sorted_alphanum([os.path.join(path, f) for f in os.listdir(path)
                            if os.path.isdir(os.path.join(path, f))])
