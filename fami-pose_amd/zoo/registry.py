"""Name -> class registries: the front door the reference's engine uses.

Mirrors utils/utils_registry.py:14-76 and engine/defaults/constant.py:9-11
(same behaviour: `@MODEL_REGISTRY.register()` decorator or call form, duplicate
names assert, unknown names raise KeyError)."""

TRAIN_PHASE, VAL_PHASE, TEST_PHASE = 'train', 'validate', 'test'


class Registry(object):
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, \
            "An object named '{}' was already registered in '{}' registry!".format(name, self._name)
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(cls):
                self._do_register(cls.__name__, cls)
                return cls
            return deco
        self._do_register(obj.__name__, obj)

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return ret


MODEL_REGISTRY = Registry("MODEL")
CORE_FUNCTION_REGISTRY = Registry("CORE_FUNCTION")
DATASET_REGISTRY = Registry("DATASET")
