"""CLI driver (mirror of ``baler/baler.py`` for the train / compress / decompress / info modes).

``python -m baler_amd --project WORKSPACE PROJECT --mode {newProject,train,compress,decompress,info}``
works on an unmodified reference-style workspace tree and writes byte-compatible artefacts
(model.pt, compressed.npz, decompressed.npz, loss_data.npy, normalization_features.npy,
activations.npy), so the reference's plot / diagnose modes can consume them.
Under ``torchrun`` (one process per GPU) training is data-parallel and compress/decompress shard rows;
rank 0 writes the artefacts.
"""
import os
import time
from math import ceil

import numpy as np

from . import dist as bdist
from .modules import helper

__all__ = ("perform_compression", "perform_decompression", "perform_training", "print_info")


def main(argv=None):
    config, mode, workspace_name, project_name, verbose = helper.get_arguments(argv)
    project_path = os.path.join("workspaces", workspace_name, project_name)
    output_path = os.path.join(project_path, "output")
    bdist.init_from_env()
    if mode == "newProject":
        helper.create_new_project(workspace_name, project_name, verbose)
    elif mode == "train":
        perform_training(output_path=output_path, config=config, verbose=verbose)
    elif mode == "compress":
        perform_compression(output_path, config, verbose)
    elif mode == "decompress":
        perform_decompression(output_path, config, verbose)
    elif mode == "info":
        print_info(output_path, config)
    elif mode in ("plot", "diagnose", "convert_with_hls4ml"):
        raise NameError(f"Baler mode {mode} consumes artefacts only; run it with the reference CLI on the "
                        "outputs written by baler_amd (the artefact formats are identical).")
    else:
        raise NameError("Baler mode " + mode + " not recognised. Use baler_amd --help to see available modes.")
    bdist.barrier()      # rank 0 writes the artefacts: no rank leaves a mode before they are complete


def perform_training(output_path, config, verbose: bool):
    """reference baler.py:84-207."""
    train_set_norm, test_set_norm, normalization_features, original_shape = helper.process(
        config.input_path, config.custom_norm, config.test_size, config.apply_normalization,
        config.convert_to_blocks if hasattr(config, "convert_to_blocks") else None, verbose,
        batch_size=bdist.global_batch(config))      # data parallel: each rank keeps only its slice of every global batch
    if verbose:
        print("Training and testing sets normalized")

    if config.data_dimension == 1:
        number_of_columns = train_set_norm.shape[1]
        config.latent_space_size = ceil(number_of_columns / config.compression_ratio)
        config.number_of_columns = number_of_columns
        n_features = number_of_columns
    elif config.data_dimension == 2:
        if getattr(config, "model_type", None) != "dense":
            raise NotImplementedError("baler_amd covers the dense models; convolutional models are out of scope")
        number_of_rows = train_set_norm.shape[1]
        number_of_columns = train_set_norm.shape[2]
        n_features = number_of_columns * number_of_rows
        config.latent_space_size = ceil((number_of_rows * number_of_columns) / config.compression_ratio)
        config.number_of_columns = number_of_columns
    else:
        raise NameError("Data dimension can only be 1 or 2. Got config.data_dimension value = "
                        + str(config.data_dimension))
    if verbose:
        print(f"Intitalizing Model with Latent Size - {config.latent_space_size} and Features - {n_features}")

    device = helper.get_device()
    if verbose:
        print(f"Device used for training: {device}")
    model_object = helper.model_init(config.model_name)
    model = model_object(n_features=n_features, z_dim=config.latent_space_size)
    model.to(device)
    if verbose:
        print(f"Model architecture:\n{model.type}")

    training_path = os.path.join(output_path, "training")
    trained_model = helper.train(model, number_of_columns, train_set_norm, test_set_norm, training_path, config)
    if verbose:
        print("Training complete")

    rank, _ = bdist.rank_world()
    if rank != 0:
        return
    if config.apply_normalization:
        np.save(os.path.join(training_path, "normalization_features.npy"), normalization_features)
    if getattr(config, "separate_model_saving", False):
        raise NotImplementedError("separate_model_saving needs model.encoder/.decoder, which the dense "
                                  "reference models do not have either (data_processing.py:60,73)")
    helper.model_saver(trained_model, os.path.join(output_path, "compressed_output", "model.pt"))
    if verbose:
        print(f"Model saved to {os.path.join(output_path, 'compressed_output', 'model.pt')}")


def perform_compression(output_path, config, verbose: bool):
    """reference baler.py:239-338."""
    print("Compressing...")
    start = time.time()
    normalization_features = []
    if config.apply_normalization:
        normalization_features = np.load(os.path.join(output_path, "training", "normalization_features.npy"))
    compressed, error_bound_batch, error_bound_deltas, error_bound_index = helper.compress(
        model_path=os.path.join(output_path, "compressed_output", "model.pt"), config=config)
    end = time.time()
    print("Compression took:", f"{(end - start) / 60:.3} minutes")
    rank, _ = bdist.rank_world()
    if rank != 0:
        return
    names = np.load(config.input_path)["names"]
    saver = np.savez_compressed if config.extra_compression else np.savez
    saver(os.path.join(output_path, "compressed_output", "compressed.npz"), data=compressed, names=names,
          normalization_features=normalization_features)
    if getattr(config, "save_error_bounded_deltas", False):    # baler.py:316-338
        helper.save_deltas(os.path.join(output_path, "compressed_output"), error_bound_batch, error_bound_deltas,
                           error_bound_index)


def perform_decompression(output_path, config, verbose: bool):
    """reference baler.py:341-456: decode, un-normalise with the TRAINING features, cast "int" columns."""
    print("Decompressing...")
    start = time.time()
    data_before_shape = helper.npz_array_shape(config.input_path, "data")   # header only: the table is not re-read
    comp_dir = os.path.join(output_path, "compressed_output")
    int_mask = None
    type_list = getattr(config, "type_list", None)
    if type_list is not None:
        # astype(int) written back into the float64 array == truncation toward zero (baler.py:426-435)
        int_mask = np.array([np.issubdtype(np.dtype(t), np.integer) for t in type_list], dtype=np.uint8)
    renorm = None
    if config.apply_normalization:   # un-normalise with the TRAINING features (baler.py:410-418), on the device
        normalization_features = np.load(os.path.join(output_path, "training", "normalization_features.npy"))
        renorm = (normalization_features, int_mask)
    decompressed, names, normalization_features = helper.decompress(
        model_path=os.path.join(comp_dir, "model.pt"),
        input_path=os.path.join(comp_dir, "compressed.npz"),
        input_path_deltas=os.path.join(comp_dir, "compressed_deltas.npz.gz"),
        input_batch_index=os.path.join(comp_dir, "compressed_batch_index_metadata.npz.gz"),
        model_name=config.model_name, config=config, output_path=output_path,
        original_shape=data_before_shape, renorm=renorm)
    rank, _ = bdist.rank_world()
    if rank != 0:
        return
    if hasattr(config, "convert_to_blocks") and config.convert_to_blocks:
        decompressed = decompressed.reshape(data_before_shape[0], data_before_shape[1], data_before_shape[2])
    if config.apply_normalization:
        print("Un-normalizing...")
    elif int_mask is not None and int_mask.any():
        decompressed = np.array(decompressed, copy=True)
        cols = np.nonzero(int_mask)[0]
        decompressed[..., cols] = np.trunc(decompressed[..., cols])
    end = time.time()
    print("Decompression took:", f"{(end - start) / 60:.3} minutes")
    saver = np.savez_compressed if config.extra_compression else np.savez
    saver(os.path.join(output_path, "decompressed_output", "decompressed.npz"), data=decompressed, names=names)


def print_info(output_path, config):
    """reference baler.py:459-510 (file-size report)."""
    comp = os.path.join(output_path, "compressed_output")
    train_dir = os.path.join(output_path, "training")
    meta = [os.path.join(comp, "model.pt"), os.path.join(train_dir, "loss_data.npy"),
            os.path.join(train_dir, "normalization_features.npy")]
    files = [config.input_path, os.path.join(comp, "compressed.npz"),
             os.path.join(output_path, "decompressed_output", "decompressed.npz")]
    mb = lambda p: os.stat(p).st_size / (1024 * 1024)  # noqa: E731
    meta_mb = sum(mb(p) for p in meta)
    f = [mb(p) for p in files]
    print("================================== \n Information about your compression \n================================== ")
    print(f"\nCompressed file is {round(f[1] / f[0], 4) * 100}% the size of the original\n")
    print(f"File size before compression: {round(f[0], 4)} MB\n")
    print(f"Compressed file size: {round(f[1], 4)} MB\n")
    print(f"De-compressed file size: {round(f[2], 4)} MB\n")
    print(f"Compression ratio: {round(f[0] / f[1], 4)}\n")
    print(f"The meta-data saved has a total size of: {round(meta_mb, 4)} MB\n")
    print(f"Combined, the actual compression ratio is: {round(f[0] / (f[1] + meta_mb), 4)}")
    print("\n ==================================")
