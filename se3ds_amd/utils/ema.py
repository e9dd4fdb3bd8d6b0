"""Exponential moving average of generator variables -- reference utils/ema.py:25-88, as two
multi-tensor launches over the flat parameter arenas (trainable + non-trainable variables:
the reference averages ALL `generator.variables`, including BN moving statistics and `u`)."""
from se3ds_amd import _lib
from se3ds_amd import hipops  # noqa: F401


def assign_ema_vars_from_initial_values(ema_model, model):
  """ema_var.assign(value) for every variable (reference :25-51).  Replicas hold identical
  values, so the cross-replica MEAN of the reference is the identity."""
  ema_model.store.theta.copy_(model.store.theta)
  ema_model.store.state.copy_(model.store.state)
  ema_model.store.version += 1


def update_ema_variables(ema_model, model, ema_decay, skip_trainable=False):
  """ema_var -= (1 - ema_decay) * (ema_var - var)  (reference :54-88).  skip_trainable: the
  optimiser pass already advanced the trainable arena (se3ds_multi_adam_keras_ema)."""
  omd = 1.0 - ema_decay
  L = _lib.lib()
  pairs = ((ema_model.store.theta, model.store.theta), (ema_model.store.state, model.store.state))
  for e, v in pairs[1:] if skip_trainable else pairs:
    _lib.check(L.se3ds_multi_ema(e.data_ptr(), v.data_ptr(), e.numel(), omd, _lib.stream()),
               'se3ds_multi_ema')
  ema_model.store.version += 1
