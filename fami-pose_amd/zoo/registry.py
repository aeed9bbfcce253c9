"""Name -> class registries: the front door the reference's engine uses.

Same observable behaviour as utils/utils_registry.py:14-76 and the three instances of
engine/defaults/constant.py:9-11: `@REG.register()` decorator or `REG.register(cls)` call form keyed by
`cls.__name__`; registering a name twice is an AssertionError; looking up an unknown name is a KeyError
(messages kept, callers and logs match on them)."""

TRAIN_PHASE, VAL_PHASE, TEST_PHASE = 'train', 'validate', 'test'


class Registry(dict):
    """A dict of classes that knows its own label."""

    def __init__(self, label):
        super().__init__()
        self.label = label

    def _add(self, cls):
        key = cls.__name__
        assert key not in self, "An object named '{}' was already registered in '{}' registry!".format(key, self.label)
        self[key] = cls
        return cls

    def register(self, cls=None):
        if cls is None:
            return self._add          # decorator form
        self._add(cls)

    def get(self, key, default=None):
        if key not in self:
            raise KeyError("No object named '{}' found in '{}' registry!".format(key, self.label))
        return self[key]


MODEL_REGISTRY = Registry("MODEL")
CORE_FUNCTION_REGISTRY = Registry("CORE_FUNCTION")
DATASET_REGISTRY = Registry("DATASET")
