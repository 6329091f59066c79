"""Layer addressing by dotted name (reference: dlmc/utils/access.py:12-61)."""
import re
from operator import attrgetter
from typing import Iterable, List

from torch import nn

__all__ = ["attrsetter", "get_layers", "mark_modules"]


def attrsetter(*paths):
    """attrsetter("a.b.c")(obj, v)  ==  obj.a.b.c = v  (for every path given)."""

    def assign(obj, value):
        for path in paths:
            *parents, leaf = path.split(".")
            holder = obj
            for name in parents:
                holder = getattr(holder, name)
            setattr(holder, leaf, value)

    return assign


def get_layers(model: nn.Module, filter_regexp: str = "(.*?)", filter_types: Iterable[nn.Module] = None) -> List[str]:
    """Names of the layers that own a weight, optionally filtered by a regular expression (matched at
    the start of the name, with an optional DataParallel `module.` prefix) and by module type."""
    names, seen = [], set()
    for pname, _ in model.named_parameters():
        if "bias" in pname:
            continue
        lname = pname.replace(".weight_orig", "").replace(".weight", "")
        if lname not in seen:
            seen.add(lname)
            names.append(lname)
    pattern = re.compile("(module\\.)?" + "(" + filter_regexp + ")")
    names = [n for n in names if pattern.match(n)]
    if filter_types is not None:
        filter_types = tuple(filter_types)
        names = [n for n in names if isinstance(_resolve(model, n), filter_types)]
    return names


def _resolve(model, name):
    try:
        return attrgetter(name)(model)
    except AttributeError:
        return None


def mark_modules(module: nn.Module, recursion: bool = True, ancestors: List[str] = ()) -> None:
    """Give every child module a `.name` attribute holding its dotted path."""
    for name, child in module._modules.items():
        path = list(ancestors) + [name]
        child.name = ".".join(path)
        if recursion:
            mark_modules(child, recursion, path)
