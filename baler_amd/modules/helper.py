"""Facade of the CLI modes (mirror of ``baler/modules/helper.py`` for train/compress/decompress).

Same function names, argument meaning and return shapes as the reference so ``baler_amd.baler`` reads
like ``baler.baler``; the loops inside ``compress`` / ``decompress`` are replaced by ONE native call
per large row block with a pre-allocated output (the reference grows a numpy array by
``np.concatenate`` per 512-row batch, helper.py:608-611,720-723 -- quadratic copying).
"""
import argparse
import importlib
import os
import sys
from dataclasses import dataclass
from math import ceil

import numpy as np
import torch

sys.path.append(os.getcwd())

from .. import dist as bdist
from .. import native
from . import data_processing, training

# rows per native encode/decode call when streaming a dataset through the device
ROW_BLOCK = 1 << 22


def get_arguments(argv=None):
    """reference helper.py:34-101: --mode / --project WORKSPACE PROJECT / --verbose, then
    ``workspaces.<W>.<P>.config.<P>_config.set_config(Config)``."""
    parser = argparse.ArgumentParser(
        prog="baler_amd",
        description="MI355X-native train/compress/decompress path of Baler (same CLI as `baler`).",
        formatter_class=argparse.RawTextHelpFormatter)
    parser.add_argument("--mode", type=str, required=True,
                        help="newProject, train, compress, decompress, info")
    parser.add_argument("--project", type=str, required=True, nargs=2, metavar=("WORKSPACE", "PROJECT"),
                        help="Specifies workspace and project, e.g. --project CMS_workspace CMS_project_v1")
    parser.add_argument("--verbose", dest="verbose", action="store_true", help="Verbose mode")
    parser.set_defaults(verbose=False)
    args = parser.parse_args(argv)

    workspace_name, project_name = args.project
    config_path = f"workspaces.{workspace_name}.{project_name}.config.{project_name}_config"
    if args.mode == "newProject":
        config = None
    else:
        config = Config
        importlib.import_module(config_path).set_config(config)
    return config, args.mode, workspace_name, project_name, args.verbose


def create_new_project(workspace_name: str, project_name: str, verbose: bool = False,
                       base_path: str = "workspaces") -> None:
    """reference helper.py:104-147: directory tree + default config."""
    workspace_path = os.path.join(base_path, workspace_name)
    project_path = os.path.join(base_path, workspace_name, project_name)
    if os.path.exists(project_path):
        print(f"The workspace and project ({project_path}) already exists.")
        return
    os.makedirs(project_path)
    required = [
        os.path.join(workspace_path, "data"),
        os.path.join(project_path, "config"),
        os.path.join(project_path, "output", "compressed_output"),
        os.path.join(project_path, "output", "decompressed_output"),
        os.path.join(project_path, "output", "plotting"),
        os.path.join(project_path, "output", "training"),
    ]
    if verbose:
        print(f"Creating project {project_name} in workspace {workspace_name}...")
    for d in required:
        if verbose:
            print(f"Creating directory {d}...")
        os.makedirs(d, exist_ok=True)
    with open(os.path.join(project_path, "config", f"{project_name}_config.py"), "w") as f:
        f.write(create_default_config(workspace_name, project_name))


@dataclass
class Config:
    """Mutable configuration holder: ``set_config(c)`` assigns attributes onto the CLASS, exactly as the
    reference does (helper.py:92-93,150-179), so config modules are interchangeable."""
    input_path: str
    compression_ratio: float
    epochs: int
    early_stopping: bool
    lr_scheduler: bool
    lr_scheduler_patience: int
    min_delta: int
    model_name: str
    custom_norm: bool
    lr: float
    batch_size: int
    test_size: float
    data_dimension: int


def create_default_config(workspace_name: str, project_name: str) -> str:
    """Default project config with the reference's keys (helper.py:182-232)."""
    lines = [
        ("input_path", f'"workspaces/{workspace_name}/data/{project_name}_data.npz"'),
        ("data_dimension", "1"), ("compression_ratio", "2.0"), ("apply_normalization", "True"),
        ("model_name", '"AE"'), ("model_type", '"dense"'), ("epochs", "5"), ("lr", "0.001"),
        ("batch_size", "512"), ("early_stopping", "True"), ("lr_scheduler", "True"),
        ("early_stopping_patience", "100"), ("min_delta", "0"), ("lr_scheduler_patience", "50"),
        ("custom_norm", "False"), ("reg_param", "0.001"), ("RHO", "0.05"), ("test_size", "0"),
        ("extra_compression", "False"), ("intermittent_model_saving", "False"),
        ("intermittent_saving_patience", "100"), ("mse_avg", "False"), ("mse_sum", "True"),
        ("emd", "False"), ("l1", "True"), ("activation_extraction", "False"),
        ("deterministic_algorithm", "True"), ("separate_model_saving", "False"),
        ("save_error_bounded_deltas", "False"), ("error_bounded_requirement", "10"),
        ("convert_to_blocks", "False"),
    ]
    body = "\n".join(f"    c.{k:<30} = {v}" for k, v in lines)
    return f"\n# === Configuration options ===\n\ndef set_config(c):\n{body}\n"


def model_init(model_name: str):
    return data_processing.initialise_model(model_name)


def numpy_to_tensor(data):
    return torch.from_numpy(data)


def get_device():
    """reference helper.py:425-439 returns "cuda:0" or "cpu"; here the device is the rank's GPU and a
    missing GPU is an error (no CPU fallback for the hot path)."""
    native.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def detacher(tensor):
    return tensor.cpu().detach().numpy()


def normalize(data, custom_norm):
    """reference helper.py:261-274 (per-column min-max over axis 0; identity if custom_norm)."""
    return data_processing.normalize(data, custom_norm)


def renormalize(data, true_min_list, feature_range_list, int_mask=None):
    return data_processing.renormalize_func(data, true_min_list, feature_range_list, int_mask)


def npz_array_shape(path, key):
    """Shape of one array of an .npz archive from its .npy header (np.load(path)[key].shape reads the whole array)."""
    import zipfile
    with zipfile.ZipFile(path) as zf, zf.open(key + ".npy") as f:
        version = np.lib.format.read_magic(f)
        header = np.lib.format.read_array_header_1_0(f) if version == (1, 0) else np.lib.format.read_array_header_2_0(f)
        return tuple(header[0])


def _load_to_device(path):
    loaded = np.load(path)
    data = loaded["data"]
    t = torch.from_numpy(np.ascontiguousarray(data))
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    return t.to(get_device()), data.shape


def process(input_path, custom_norm, test_size, apply_normalization, convert_to_blocks, verbose):
    """reference helper.py:277-319.  Returns (train_set, test_set, normalization_features,
    original_shape); the two sets are DEVICE tensors (the dataset crosses PCIe once), the features a
    numpy array as in the reference."""
    data, original_shape = _load_to_device(input_path)
    if verbose:
        print("Original Dataset Shape - ", tuple(original_shape))
    if convert_to_blocks:
        data = data_processing.convert_to_blocks_util(convert_to_blocks, data)
    feats = data_processing.find_minmax(data)
    normalization_features = feats.cpu().numpy()
    if apply_normalization:
        print("Normalizing the data...")
        if not custom_norm:
            flat = data.reshape(data.shape[0], -1)
            data = native.normalize(flat.contiguous(), feats.reshape(2, -1).contiguous(),
                                    torch.float64).reshape(data.shape)
    if not test_size:
        train_set = data
        test_set = train_set
    else:
        train_set, test_set = data_processing.split(data, test_size=test_size, random_state=1)
    return train_set, test_set, normalization_features, original_shape


def train(model, number_of_columns, train_set, test_set, project_path, config):
    return training.train(model, number_of_columns, train_set, test_set, project_path, config)


def model_saver(model, model_path):
    return data_processing.save_model(model, model_path)


def _derive_sizes(config, data_shape, names_len):
    """n_features / latent size exactly as helper.compress derives them (helper.py:505-535)."""
    if config.data_dimension == 1:
        number_of_columns = names_len
        config.latent_space_size = ceil(number_of_columns / config.compression_ratio)
        config.number_of_columns = number_of_columns
        return number_of_columns
    if config.data_dimension == 2:
        if getattr(config, "model_type", None) != "dense":
            raise NotImplementedError("baler_amd covers the dense models; convolutional models are out of scope")
        number_of_rows = data_shape[1]
        config.number_of_columns = data_shape[2]
        config.latent_space_size = ceil((number_of_rows * config.number_of_columns) / config.compression_ratio)
        return number_of_rows * config.number_of_columns
    raise NameError("Data dimension can only be 1 or 2. Got config.data_dimension = "
                    + str(config.data_dimension))


def compress(model_path, config):
    """reference helper.py:473-616.  Returns (compressed ndarray, batches, deltas, indices); the last three are
    empty lists unless ``config.save_error_bounded_deltas`` (then: batch numbers, one float16 array per batch and
    one ``(rows, cols)`` tuple per batch, helper.py:589-606).  The input is re-normalised with ITS OWN min/max
    (helper.py:500-504); with ``torch.distributed`` the rows are sharded over ranks with no collective and
    gathered in rank order."""
    want_deltas = bool(getattr(config, "save_error_bounded_deltas", False))
    data, original_shape = _load_to_device(config.input_path)
    if hasattr(config, "convert_to_blocks") and config.convert_to_blocks:
        data = data_processing.convert_to_blocks_util(config.convert_to_blocks, data)
    names = np.load(config.input_path)["names"]
    n_features = _derive_sizes(config, data.shape, len(names))
    flat = data.reshape(data.shape[0], -1).contiguous()
    feats = None
    if config.apply_normalization and not config.custom_norm:
        print("Normalizing...")
        feats = native.minmax(flat)
    if config.data_dimension == 2:
        flat = flat.to(torch.float32)  # reference: torch.tensor(data, dtype=float32) (helper.py:560-563)

    model = data_processing.load_model(data_processing.initialise_model(config.model_name), model_path,
                                       n_features=n_features, z_dim=config.latent_space_size)
    model.eval()
    h = model.handle()
    rank, world = bdist.rank_world()
    lo, hi = bdist.shard_rows(flat.shape[0], rank, world)
    out = torch.empty((hi - lo, config.latent_space_size), dtype=flat.dtype, device=flat.device)
    if want_deltas:
        flags = torch.empty((hi - lo, n_features), dtype=torch.uint8, device=flat.device)
        deltas = torch.empty((hi - lo, n_features), dtype=torch.float16, device=flat.device)
    for s in range(lo, hi, ROW_BLOCK):
        e = min(s + ROW_BLOCK, hi)
        if not want_deltas:
            out[s - lo:e - lo] = h.encode(flat[s:e], features=feats)
            continue
        # the side channel compares decode(encode(x)) with the NORMALISED input (helper.py:589-606)
        xn = native.normalize(flat[s:e], feats, out_dtype=flat.dtype) if feats is not None else flat[s:e]
        z = h.encode(xn)
        out[s - lo:e - lo] = z
        flags[s - lo:e - lo], deltas[s - lo:e - lo] = native.error_deltas(xn, h.decode(z),
                                                                          config.error_bounded_requirement)
    compressed = _gather_rows(out, flat.shape[0], world)
    if not want_deltas:
        return compressed, [], [], []
    flags = _gather_rows(flags, flat.shape[0], world)
    deltas = _gather_rows(deltas, flat.shape[0], world)
    return (compressed,) + split_deltas(flags, deltas, config.batch_size)


def save_error_bounded_requirement(config, decoded_output, data_batch):
    """reference helper.py:442-470 for ONE batch of host arrays (kept for callers of the reference function; the
    compress path evaluates the whole table at once): -> (list of float16 deltas, (rows, cols))."""
    dev = get_device()
    x = torch.as_tensor(np.ascontiguousarray(data_batch), device=dev)
    r = torch.as_tensor(np.ascontiguousarray(decoded_output), device=dev).to(x.dtype)
    flags, deltas = native.error_deltas(x, r, config.error_bounded_requirement)
    rows, cols = np.nonzero(flags.cpu().numpy())
    return list(deltas.cpu().numpy()[rows, cols]), (rows, cols)


def split_deltas(flags, deltas, batch_size):
    """Dense (flags, float16 deltas) of the whole table -> the reference's per-batch side channel
    (helper.py:589-606): batch numbers, flagged deltas (row-major, as ``np.where`` orders them) and
    ``(rows, cols)`` with rows counted inside the batch.  Every batch is listed, flagged or not."""
    batches, out_deltas, out_index = [], [], []
    for idx, s in enumerate(range(0, flags.shape[0], batch_size)):
        rows, cols = np.nonzero(flags[s:s + batch_size])
        batches.append(idx)
        out_deltas.append(np.ascontiguousarray(deltas[s:s + batch_size][rows, cols]))
        out_index.append((rows, cols))
    print("Total Deltas Found - ", int(sum(len(d) for d in out_deltas)))
    return batches, out_deltas, out_index


def save_deltas(comp_dir, batches, deltas, index):
    """baler.py:316-338: the two gzip'd ``np.save`` object arrays the reference's decompress reads back
    (``compressed_deltas.npz.gz``: one entry per batch; ``compressed_batch_index_metadata.npz.gz``:
    ``[batch numbers, (rows, cols) per batch]``).  Entries are float16 ARRAYS where the reference stored Python
    lists of float16 scalars -- the consumer indexes them the same way (helper.py:708-718)."""
    import gzip
    d_arr = np.empty(len(batches), dtype=object)
    i_arr = np.empty((2, len(batches)), dtype=object)
    for k, b in enumerate(batches):
        d_arr[k] = deltas[k]
        i_arr[0, k] = b
        i_arr[1, k] = index[k]
    with gzip.GzipFile(os.path.join(comp_dir, "compressed_deltas.npz.gz"), "w") as f:
        np.save(file=f, arr=d_arr)
    with gzip.GzipFile(os.path.join(comp_dir, "compressed_batch_index_metadata.npz.gz"), "w") as f:
        np.save(file=f, arr=i_arr)


def load_deltas(input_path_deltas, input_batch_index, batch_size):
    """Side channel files -> flat (global rows int64, cols int32, deltas float16) host arrays (helper.py:655-665)."""
    import gzip
    loaded_deltas = np.load(gzip.GzipFile(input_path_deltas, "r"), allow_pickle=True)
    loaded_index = np.load(gzip.GzipFile(input_batch_index, "r"), allow_pickle=True)
    rows, cols, vals = [], [], []
    for k, b in enumerate(loaded_index[0]):
        r, c = loaded_index[1][k]
        rows.append(np.asarray(r, dtype=np.int64) + int(b) * batch_size)
        cols.append(np.asarray(c, dtype=np.int32))
        vals.append(np.asarray(loaded_deltas[k], dtype=np.float16).reshape(-1))
    if not rows:
        return np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float16)
    return np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)


def _gather_rows(local, n_total, world):
    """Concatenate per-rank row shards in rank order on the host (no data-path collective needed for a
    single rank; multi-rank uses all_gather_object of host arrays, off the hot path)."""
    host = local.cpu().numpy()
    if world == 1:
        return host
    import torch.distributed as td
    parts = [None] * world
    td.all_gather_object(parts, host)
    return np.concatenate(parts)


def decompress(model_path, input_path, input_path_deltas, input_batch_index, model_name, config,
               output_path, original_shape, renorm=None):
    """reference helper.py:619-733.  Returns (decompressed ndarray, names, normalization_features).  With
    ``config.save_error_bounded_deltas`` the stored float16 deltas are subtracted from the decoder output of
    their batch (helper.py:708-718), on the device, before un-normalisation.

    ``renorm`` (not in the reference signature; optional): ``(features (2, C) float64, int_mask or None)``.  When
    given, the un-normalisation ``x*range + min`` and the truncation of the "int" columns that the reference applies
    afterwards on the host (baler.py:410-435) run on the device -- fused into the decode kernel's store when there are
    no deltas -- so the decompressed table crosses PCIe once (float64) instead of three times."""
    loaded = np.load(input_path)
    data = loaded["data"]
    names = loaded["names"]
    normalization_features = loaded["normalization_features"]
    latent_space_size = len(data[0])
    model_dict = torch.load(str(model_path), map_location="cpu")
    number_of_columns = len(model_dict[list(model_dict.keys())[-1]])  # len(de4.bias), helper.py:668-674

    model = data_processing.load_model(data_processing.initialise_model(config.model_name), model_path,
                                       n_features=number_of_columns, z_dim=latent_space_size)
    model.eval()
    h = model.handle()
    z = torch.from_numpy(np.ascontiguousarray(data))
    if z.dtype not in (torch.float32, torch.float64):
        z = z.to(torch.float64)
    z = z.to(get_device())
    rank, world = bdist.rank_world()
    lo, hi = bdist.shard_rows(z.shape[0], rank, world)
    want_deltas = bool(getattr(config, "save_error_bounded_deltas", False))
    r_feats = r_mask = None
    if renorm is not None:
        r_feats = torch.as_tensor(np.asarray(renorm[0], dtype=np.float64).reshape(2, -1), device=z.device).contiguous()
        if renorm[1] is not None:
            r_mask = torch.as_tensor(np.asarray(renorm[1], dtype=np.uint8), device=z.device).contiguous()
    fuse = renorm is not None and not want_deltas
    out = torch.empty((hi - lo, number_of_columns), dtype=torch.float64 if fuse else z.dtype, device=z.device)
    for s in range(lo, hi, ROW_BLOCK):
        e = min(s + ROW_BLOCK, hi)
        out[s - lo:e - lo] = (h.decode(z[s:e], features=r_feats, int_mask=r_mask, out_dtype=torch.float64) if fuse
                              else h.decode(z[s:e]))
    if want_deltas:
        rows, cols, vals = load_deltas(input_path_deltas, input_batch_index, config.batch_size)
        mine = (rows >= lo) & (rows < hi)                     # this rank's row shard
        dev = out.device
        native.apply_deltas(out, torch.from_numpy(rows[mine] - lo).to(dev), torch.from_numpy(cols[mine]).to(dev),
                            torch.from_numpy(vals[mine]).to(dev))
        print("Total Deltas Added - ", int(len(rows)))
        if renorm is not None:
            out = native.renormalize(out, r_feats, r_mask)
    decompressed = _gather_rows(out, z.shape[0], world)
    if config.data_dimension == 2 and getattr(config, "model_type", None) == "dense":
        blocks = getattr(config, "convert_to_blocks", None)
        if blocks:
            # rows are blocks here, not frames.  The reference reshapes with the FRAME shape (helper.py:728-731) and so
            # raises ValueError for its own exafel1/exafel2 configs (dense model + convert_to_blocks); the caller
            # (perform_decompression, baler.py:394-408) folds the blocks back into frames.
            decompressed = decompressed.reshape((len(decompressed), blocks[1], blocks[2]))
        else:
            decompressed = decompressed.reshape((len(decompressed), original_shape[1], original_shape[2]))
    return decompressed, names, normalization_features
