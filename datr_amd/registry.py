"""Name -> build-function registry: `MODULE_BUILD_FUNCS.get('dino')(args)` returns
(model, criterion, postprocessors), which is how the reference's driver builds the model
(/root/reference/main.py:79-85; the reference's registry: /root/reference/models/registry.py:12-57).
Only the surface the path uses: the decorator that registers a build function under a name and `get`."""
from __future__ import annotations

from typing import Callable, Dict, Optional


class Registry:
    def __init__(self, name: str):
        self.name = name
        self._builders: Dict[str, Callable] = {}

    def get(self, key: str) -> Optional[Callable]:
        return self._builders.get(key)

    def registe_with_name(self, module_name: Optional[str] = None, force: bool = False):
        """Decorator (the reference's spelling, kept because model files use it):
        `@MODULE_BUILD_FUNCS.registe_with_name(module_name='dino')`."""
        def decorate(fn: Callable) -> Callable:
            if not callable(fn):
                raise TypeError(f"a build function is expected, got {type(fn)}")
            key = module_name or fn.__name__
            if key in self._builders and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._builders[key] = fn
            return fn
        return decorate


MODULE_BUILD_FUNCS = Registry("model build functions")
