"""A small gin-config subset: exactly the grammar the SE3DS configs use
(configs/highres/highres.gin, configs/lowres/lowres.gin and the trainer test's config block,
reference trainers/se3ds_trainer_test.py:70-99):

    Selector.param = literal            # numbers, strings, True/False/None, lists, tuples
    Selector.param = @module.Class      # reference to another configurable
    # comments, blank lines

Resolution order when a configurable is called: explicit kwargs > binding > Python default.
Selectors match a configurable by its (possibly module-qualified) name suffix, e.g.
`image_models.ResNetGenerator.gen_dims` and `ResNetGenerator.gen_dims` both bind
`se3ds_amd.models.image_models.ResNetGenerator`.  Bindings for unknown configurables (dataset
classes, inception_model, ...) are kept but never consumed, as gin does for unused modules.
"""
import ast
import functools
import inspect
from typing import Any, Dict

_REGISTRY: Dict[str, Any] = {}            # 'module.Class' -> wrapped callable
_BINDINGS: Dict[str, Dict[str, Any]] = {}  # selector (as written) -> {param: value}
_DENYLIST: Dict[str, set] = {}


class _Ref:
  def __init__(self, selector):
    self.selector = selector

  def resolve(self):
    fn = _lookup(self.selector)
    if fn is None:
      raise ValueError(f"No configurable matching reference '@{self.selector}'.")
    return fn


def _lookup(selector):
  hits = [v for k, v in _REGISTRY.items() if k == selector or k.endswith('.' + selector)]
  if not hits:
    return None
  if len(set(map(id, hits))) > 1:
    raise ValueError(f"Ambiguous selector '{selector}'.")
  return hits[0]


def _bindings_for(full_name):
  out = {}
  for sel, params in _BINDINGS.items():
    if full_name == sel or full_name.endswith('.' + sel):
      out.update(params)
  return out


def configurable(arg=None, *, denylist=None, module=None):
  """Decorator: registers a class / function as gin-configurable."""
  def wrap(fn):
    mod = module or fn.__module__.split('.')[-1]
    full = f'{mod}.{fn.__name__}'
    _DENYLIST[full] = set(denylist or ())
    if inspect.isclass(fn):
      orig_init = fn.__init__
      sig = inspect.signature(orig_init)
      accepts_kwargs = any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values())
      names = [p for p in sig.parameters if p != 'self']

      @functools.wraps(orig_init)
      def init(self, *args, **kwargs):
        # bindings written for a subclass selector reach base-class parameters through
        # **kwargs (reference configs bind `se3ds_trainer.GAN.predict_depth`, a GANManager arg)
        own = f'{mod}.{type(self).__name__}' if type(self).__init__ is init else full
        bound = dict(_bindings_for(full))
        if own != full:
          bound.update(_bindings_for(own))
        positional = set(names[:len(args)])
        for k, v in bound.items():
          if k in kwargs or k in positional or k in _DENYLIST[full]:
            continue
          if k in names or accepts_kwargs:
            kwargs[k] = v.resolve() if isinstance(v, _Ref) else v
        orig_init(self, *args, **kwargs)

      fn.__init__ = init
      _REGISTRY[full] = fn
      return fn

    sig = inspect.signature(fn)
    names = list(sig.parameters)

    @functools.wraps(fn)
    def call(*args, **kwargs):
      positional = set(names[:len(args)])
      for k, v in _bindings_for(full).items():
        if k not in kwargs and k not in positional and k in names:
          kwargs[k] = v.resolve() if isinstance(v, _Ref) else v
      return fn(*args, **kwargs)

    _REGISTRY[full] = call
    return call

  if callable(arg):
    return wrap(arg)
  return wrap


def _parse_value(text):
  text = text.strip()
  if text.startswith('@'):
    return _Ref(text[1:].rstrip('()'))
  if text.startswith('%'):
    raise ValueError('gin macros are not supported by gin_lite')
  return ast.literal_eval(text)


def parse_config(config: str):
  """Parses bindings from a string (one `selector.param = value` per logical line)."""
  def strip_comment(raw):
    quote = None
    for i, ch in enumerate(raw):
      if quote:
        if ch == quote:
          quote = None
      elif ch in '\'"':
        quote = ch
      elif ch == '#':
        return raw[:i]
    return raw

  pending = ''
  for raw in config.splitlines():
    line = strip_comment(raw).rstrip()
    if not line.strip():
      continue
    pending += line
    if pending.count('(') > pending.count(')') or pending.count('[') > pending.count(']'):
      continue
    if '=' not in pending:
      raise ValueError(f'gin_lite: cannot parse line {pending!r}')
    lhs, rhs = pending.split('=', 1)
    pending = ''
    lhs = lhs.strip()
    if '/' in lhs:
      lhs = lhs.split('/')[-1]   # drop scopes
    selector, param = lhs.rsplit('.', 1)
    rhs = rhs.strip()
    _BINDINGS.setdefault(selector, {})[param] = _parse_value(rhs)


def parse_config_files_and_bindings(config_files=None, bindings=None):
  """gin.parse_config_files_and_bindings (reference main.py:47)."""
  for f in ([config_files] if isinstance(config_files, str) else (config_files or [])):
    with open(f) as fh:
      parse_config(fh.read())
  for b in ([bindings] if isinstance(bindings, str) else (bindings or [])):
    parse_config(b)


def bind_parameter(binding_key: str, value):
  selector, param = binding_key.rsplit('.', 1)
  _BINDINGS.setdefault(selector, {})[param] = value


def query_parameter(binding_key: str):
  selector, param = binding_key.rsplit('.', 1)
  for sel, params in _BINDINGS.items():
    if (sel == selector or sel.endswith('.' + selector) or selector.endswith('.' + sel)) and \
        param in params:
      v = params[param]
      return v.resolve() if isinstance(v, _Ref) else v
  raise ValueError(f"No binding for '{binding_key}'.")


def clear_config():
  _BINDINGS.clear()


def operative_bindings():
  return {k: dict(v) for k, v in _BINDINGS.items()}
