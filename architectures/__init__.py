"""Drop-in mirror of the reference's ``architectures`` package for the segmentor+discriminator
training hot path, running on libocta_hip.so (MI355X).  Same module paths, class names,
constructor signatures, attribute and ``state_dict`` key names as IoBT-VISTEC/OCTAve;
see INTEGRATION.md.  Components of the reference that are off the hot path (parallel-head
U-Nets, CE-Net ResNet, propagation/aggregation nets) are not provided.
"""
