"""Error-message helpers with the reference's exact wording.

Mirrors torch_nf/error_formatters.py:4-34 of the reference: the TypeError text
"<Class> argument <name> must be <type> not <type>." is part of the drop-in
contract (the reference's tests match on exception types and these strings).
"""
import torch


def format_type_err_msg(obj, arg_name, arg, correct_type):
    """Text of the TypeError raised when `arg` (named `arg_name`) handed to `obj`
    is not exactly `correct_type` (error_formatters.py:4-24)."""
    got = arg.__class__
    if got is correct_type:
        raise ValueError("Invalid TypeError message: type(arg) == correct_type.")
    return "{} argument {} must be {} not {}.".format(
        obj.__class__.__name__, arg_name, correct_type.__name__, got.__name__
    )


def dbg_check(tensor, name):
    """Print and return the inf/nan census of a tensor (error_formatters.py:26-34)."""
    total = tensor.numel()
    n_inf = int(torch.isinf(tensor).sum().item())
    n_nan = int(torch.isnan(tensor).sum().item())
    print(name, "infs %d/%d" % (n_inf, total), "nans %d/%d" % (n_nan, total))
    return n_nan or n_inf
