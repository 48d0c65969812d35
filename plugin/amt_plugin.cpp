// amt_plugin.cpp -- the AviSynth plugin entry point of the GPU logo path.
//
// Registers the two logo filters under the names and argument specifications the reference's Amatsukaze.dll uses
// (Amatsukaze.cpp:43-65: "AMTAnalyzeLogo" "cs[maskratio]i", "AMTEraseLogo" "ccs[logof]s[mode]i[maxfade]i"), with Create
// factories that read their arguments exactly like the reference's (LogoScan.hpp:1227-1235: maskratio is a percentage,
// default 35; :1507-1518: logof "", mode 0, maxfade 16).  FilteredSource's MakeSource script
//     AMTEraseLogo(src, AMTAnalyzeLogo(src, logo[, maskratio=..]), logo, logof=.., maxfade=..)
// therefore runs unchanged against this plugin.  The reference's other registrations (AMTSource, AMTDecimate, AMTExec,
// AMTOrderedParallel) are not pixel work and stay in Amatsukaze.dll.
//
// Host headers: on Windows / AviSynthNeo compile with -DAMT_FILTERS_USE_AVISYNTH_H against the real avisynth.h; here
// (Linux, no AviSynth) it builds against include/amt_avs_min.h so that the registration and the factories can be tested.
#include <memory>
#include <mutex>

#include "amt_filters.hpp"

#ifdef AMT_FILTERS_USE_AVISYNTH_H
const AVS_Linkage* AVS_linkage = nullptr;
#define AMT_PLUGIN_EXPORT extern "C" __declspec(dllexport)
#define AMT_STDCALL __stdcall
#define AMT_CDECL __cdecl
#else
using namespace amtavs;
#define AMT_PLUGIN_EXPORT extern "C" __attribute__((visibility("default")))
#define AMT_STDCALL
#define AMT_CDECL
#endif

namespace {

// one GPU context (device, stream, pinned staging ring) for all filters of the process; AMTGPU_DEVICE selects the device
amtgpu::PContext shared_context(IScriptEnvironment* env)
{
    static std::mutex mu;
    static std::weak_ptr<amtgpu::Context> weak;
    std::lock_guard<std::mutex> lock(mu);
    amtgpu::PContext c = weak.lock();
    if (!c) {
        const char* d = std::getenv("AMTGPU_DEVICE");
        try { c = std::make_shared<amtgpu::Context>(d ? std::atoi(d) : 0); }
        catch (const std::exception& e) { env->ThrowError("%s", e.what()); }
        weak = c;
    }
    return c;
}

// user_data: the analysis mode.  "AMTAnalyzeLogo" is the reference's filter bit for bit (AMTGPU_ANALYZE_EXACT); "AMTAnalyzeLogoFast"
// evaluates all fades from one window evaluation (AMTGPU_ANALYZE_LINEAR_GUARDED): every fade AMTEraseLogo takes from the clip --
// its only consumer, CalcFade2, LogoScan.hpp:1288-1314 -- is identical, the clip's floats are within 1e-4 of the reference's.
AVSValue AMT_CDECL Create_AMTAnalyzeLogo(AVSValue args, void* user_data, IScriptEnvironment* env)
{
    return new amtgpu::AMTAnalyzeLogo(args[0].AsClip(),                               // source
                                      args[1].AsString(),                             // logopath
                                      (float)args[2].AsFloat(35) / 100.0f,            // maskratio (percent)
                                      env, shared_context(env), 32, user_data ? AMTGPU_ANALYZE_LINEAR_GUARDED : AMTGPU_ANALYZE_EXACT);
}

AVSValue AMT_CDECL Create_AMTEraseLogo(AVSValue args, void*, IScriptEnvironment* env)
{
    return new amtgpu::AMTEraseLogo(args[0].AsClip(),                                 // source
                                    args[1].AsClip(),                                 // analyzeclip
                                    args[2].AsString(),                               // logopath
                                    args[3].AsString(""),                             // logofpath
                                    args[4].AsInt(0),                                 // mode
                                    args[5].AsInt(16),                                // maxfade
                                    env, shared_context(env));
}

} // namespace

AMT_PLUGIN_EXPORT const char* AMT_STDCALL AvisynthPluginInit3(IScriptEnvironment* env, const AVS_Linkage* const vectors)
{
#ifdef AMT_FILTERS_USE_AVISYNTH_H
    AVS_linkage = vectors;
#else
    (void)vectors;
#endif
    env->AddFunction("AMTAnalyzeLogo", "cs[maskratio]i", Create_AMTAnalyzeLogo, 0);
    env->AddFunction("AMTEraseLogo", "ccs[logof]s[mode]i[maxfade]i", Create_AMTEraseLogo, 0);
    // not a name of the reference's: the opt-in fast analysis for scripts that want it (same arguments)
    env->AddFunction("AMTAnalyzeLogoFast", "cs[maskratio]i", Create_AMTAnalyzeLogo, (void*)1);
    return "Amatsukaze GPU logo plugin";
}
