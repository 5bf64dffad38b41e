"""``torch.amp.custom_fwd(cast_inputs=...)`` / ``custom_bwd`` with a shortcut for the case the bf16 step is in all the
time: autocast is on and every floating-point argument ALREADY has the cast dtype (bf16 weights + bf16 activations,
``Runner(bf16_params=True)``).  torch's decorators then still walk the arguments and enter an ``autocast(enabled=False)``
context on the way in and an ``autocast`` context on the way back -- ~15 us of host time per call each way, 50-100
calls per step in a ResNet trunk whose step is paced by the host.  With nothing to cast, the forward can run under the
ambient autocast (its ops see operands of the autocast dtype: no casts happen) and the backward as it is (the autograd
thread runs with autocast off, which is what ``custom_bwd`` would establish for a forward that cast its inputs).
Any other situation takes torch's own decorators, unchanged."""
import torch


def light_custom_fwd(cast_inputs):
    std = torch.amp.custom_fwd(device_type='cuda', cast_inputs=cast_inputs)

    def deco(fwd):
        slow = std(fwd)

        def wrapper(ctx, *args):
            if torch.is_autocast_enabled():
                for a in args:
                    if isinstance(a, torch.Tensor) and a.is_floating_point() and a.dtype != cast_inputs:
                        ctx._light_amp = False
                        return slow(ctx, *args)
                ctx._light_amp = True
                # what custom_fwd(cast_inputs=...) would leave on the context: if backward() is called INSIDE an autocast
                # block the wrapper below falls to torch's custom_bwd, which reads these two
                ctx._fwd_used_autocast = False
                ctx._dtype = torch.get_autocast_dtype('cuda')
                return fwd(ctx, *args)
            ctx._light_amp = False
            return slow(ctx, *args)
        return wrapper
    return deco


def light_custom_bwd(bwd):
    slow = torch.amp.custom_bwd(device_type='cuda')(bwd)

    def wrapper(ctx, *grads):
        if getattr(ctx, "_light_amp", False) and not torch.is_autocast_enabled():
            return bwd(ctx, *grads)
        return slow(ctx, *grads)
    return wrapper
