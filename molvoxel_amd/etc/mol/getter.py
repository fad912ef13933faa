"""Channel getters: what turns an atom or a bond into a channel index or a feature row.

Same contract as the reference's `molvoxel/etc/rdkit/base.py:7-52` and `getter.py:7-46` (class names, `CHANNEL_TYPE`,
`channels`, `num_channels`, `get_type`, `get_feature`, `to_feature_getter`, `BondTypeGetter.default()`), fed with
`AtomView`/`BondView` objects (or RDKit atoms/bonds: only `GetSymbol()` / `GetBondType()` are called).
"""
from __future__ import annotations

from collections.abc import Callable, Sequence
from typing import Any

import numpy as np

from .molecule import BondType


class ChannelGetter:
    CHANNEL_TYPE = ""

    def __init__(self, channels: Sequence[str]):
        self.channels = list(channels)
        self.num_channels = len(self.channels)


class FeatureGetter(ChannelGetter):
    """`function(item, **kwargs) -> (num_channels,)` feature row per atom / bond."""

    CHANNEL_TYPE = "FEATURE"

    def __init__(self, function: Callable[..., Any], channels: Sequence[str]):
        super().__init__(channels)
        self.feature_getter = function

    def get_feature(self, item: Any, **kwargs):
        return self.feature_getter(item, **kwargs)


class TypeGetter(ChannelGetter):
    """Key -> channel index. With `unknown=True` a last channel "Unknown" takes every key outside the table;
    without it an unlisted key raises `KeyError`, as the reference's dictionary lookup does (`base.py:32-35`)."""

    CHANNEL_TYPE = "TYPE"

    def __init__(self, types: Sequence[Any], channels: Sequence[str], unknown: bool = False):
        names = list(channels) + (["Unknown"] if unknown else [])
        super().__init__(names)
        self._index = {key: i for i, key in enumerate(types)}
        self._unknown = self.num_channels - 1 if unknown else None
        self._one_hot = np.eye(self.num_channels, dtype=np.float32)

    def lookup(self, key: Any) -> int:
        if self._unknown is None:
            return self._index[key]
        return self._index.get(key, self._unknown)

    def key_of(self, item: Any) -> Any:
        """The table key of an atom / bond; subclasses say which accessor provides it."""
        return item

    def get_type(self, item: Any, **kwargs) -> int:
        return self.lookup(self.key_of(item))

    def get_feature(self, item: Any, **kwargs):
        return self._one_hot[self.get_type(item, **kwargs)]

    def to_feature_getter(self) -> FeatureGetter:
        return FeatureGetter(self.get_feature, self.channels)

    def types_of_keys(self, keys: Sequence[Any]) -> np.ndarray:
        """Vectorised form used by the point-cloud makers: one dictionary pass per distinct key."""
        keys = list(keys)
        table = {k: self.lookup(k) for k in set(keys)}
        return np.fromiter((table[k] for k in keys), dtype=np.int16, count=len(keys))


# atoms ----------------------------------------------------------------------------------------------
AtomChannelGetter = ChannelGetter


class AtomFeatureGetter(FeatureGetter):
    pass


class AtomTypeGetter(TypeGetter):
    """Element symbol -> channel (`getter.py:14-21`)."""

    def __init__(self, symbols: Sequence[str], symbol_names: Sequence[str] | None = None, unknown: bool = False):
        super().__init__(symbols, symbols if symbol_names is None else symbol_names, unknown)

    def key_of(self, atom: Any) -> str:
        return atom.GetSymbol() if hasattr(atom, "GetSymbol") else atom


# bonds ----------------------------------------------------------------------------------------------
BondChannelGetter = ChannelGetter


class BondFeatureGetter(FeatureGetter):
    pass


class BondTypeGetter(TypeGetter):
    """Bond order -> channel (`getter.py:30-46`)."""

    def __init__(self, bondtypes: Sequence[Any], bondtype_names: Sequence[str] | None = None, unknown: bool = False):
        names = [str(bt) for bt in bondtypes] if bondtype_names is None else bondtype_names
        super().__init__([self._normalise(bt) for bt in bondtypes], names, unknown)

    @staticmethod
    def _normalise(bondtype: Any) -> int:
        """BondType member, molfile integer or an RDKit BondType (by name) -> molfile integer."""
        if isinstance(bondtype, (int, np.integer)):
            return int(bondtype)
        name = str(bondtype).split(".")[-1]
        return int(BondType[name]) if name in BondType.__members__ else -1

    def key_of(self, bond: Any) -> int:
        return self._normalise(bond.GetBondType() if hasattr(bond, "GetBondType") else bond)

    @classmethod
    def default(cls) -> "BondTypeGetter":
        return cls([BondType.SINGLE, BondType.DOUBLE, BondType.TRIPLE, BondType.AROMATIC],
                   ["SingleBond", "DoubleBond", "TripleBond", "AromaticBond"])
