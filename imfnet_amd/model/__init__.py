"""Model registry -- `load_model(name)` returns the class, as model/__init__.py:16-30 of the
reference does (scripts/generate_desc.py:164)."""
import logging

from . import resunet as _resunet

MODELS = [getattr(_resunet, a) for a in dir(_resunet)
          if 'Net' in a and isinstance(getattr(_resunet, a), type)]


def load_model(name):
    """Class for a model name such as 'ResUNetBN2C'; None (after logging the options) if unknown."""
    by_name = {m.__name__: m for m in MODELS}
    if name not in by_name:
        logging.info(f'Invalid model index. You put {name}. Options are:')
        for m in MODELS:
            logging.info('\t* {}'.format(m.__name__))
        return None
    return by_name[name]
