"""Validation dataset of train.ipynb — mirror of the reference's util/ref_data_load.py, which is the same class as
util/data_load.py under another name (:8-36)."""
from .data_load import Data_load


class Ref_Data_load(Data_load):
    pass
