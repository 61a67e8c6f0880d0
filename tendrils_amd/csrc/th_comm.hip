// th_comm.hip - placeholder (filled in below in this round): RCCL communicator of a context.
