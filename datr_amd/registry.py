"""Name -> build-function registry (mirror of /root/reference/models/registry.py:12-57).
`MODULE_BUILD_FUNCS.get('dino')(args)` returns (model, criterion, postprocessors), which is
how the reference's driver builds the model (/root/reference/main.py:79-85)."""
from __future__ import annotations

import inspect
from functools import partial


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __repr__(self):
        return f"{self.__class__.__name__}(name={self._name}, items={list(self._module_dict)})"

    def __len__(self):
        return len(self._module_dict)

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def registe_with_name(self, module_name=None, force=False):     # (sic) reference spelling
        return partial(self.register, module_name=module_name, force=force)

    def register(self, module_build_function, module_name=None, force=False):
        if not inspect.isfunction(module_build_function):
            raise TypeError("module_build_function must be a function, but got "
                            f"{type(module_build_function)}")
        module_name = module_name or module_build_function.__name__
        if not force and module_name in self._module_dict:
            raise KeyError(f"{module_name} is already registered in {self.name}")
        self._module_dict[module_name] = module_build_function
        return module_build_function


MODULE_BUILD_FUNCS = Registry("model build functions")
